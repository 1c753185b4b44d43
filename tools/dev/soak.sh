#!/bin/bash
# dev: longer runs of the three training loops on synthetic data (finite losses, steady rate, no memory growth)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; local t0=$(date +%s); timeout 600 "$@" 2>&1 | grep -a "Steps\|Error\|error\|nan" | tail -4; echo "wall $(( $(date +%s) - t0 )) s"; }
run python train_gan.py configs/gan/cifar10/c10_b512.gin sndcgan --mode=contrad --aug=simclr --use_warmup --synthetic --max_steps 1500 --print_every 500 --evaluate_every 100000 --logdir /tmp/soak_c10
run python train_gan.py configs/gan/cifar10/c10_b512.gin sndcgan --mode=contrad --aug=simclr --use_warmup --synthetic --max_steps 1500 --print_every 500 --evaluate_every 100000 --graph --logdir /tmp/soak_c10g
run python train_stylegan2.py configs/gan/stylegan2/c10_style64.gin stylegan2 --mode=contrad --aug=simclr --lbd_r1 0.1 --no_lazy --synthetic --max_steps 600 --print_every 200 --evaluate_every 100000 --logdir /tmp/soak_sg32
run python train_stylegan2_contraD.py configs/gan/stylegan2/afhq_dog_style64.gin stylegan2_512 --mode=contrad --aug=simclr_hq --lbd_r1 0.5 --synthetic --batch_size 16 --max_steps 96 --print_every 32 --evaluate_every 100000 --logdir /tmp/soak_sg512
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
