#!/bin/bash
# dev: images/s of the full training loop (D-step + G-step) eager vs --graph, from the CLI's own log lines
for cfg in c10_b64 c10_b512; do for g in "" "--graph"; do
  echo "== $cfg $g"
  timeout 300 python train_gan.py configs/gan/cifar10/$cfg.gin sndcgan --mode=contrad --aug=simclr --use_warmup --synthetic \
     --max_steps 400 --print_every 100 --evaluate_every 100000 --logdir /tmp/lr_$cfg$g 2>/dev/null | grep "img/s" | tail -2
done; done
for g in "" "--graph"; do
  echo "== sg2 c10_style64 $g"
  timeout 300 python train_stylegan2_contraD.py configs/gan/stylegan2/c10_style64.gin stylegan2 --mode=contrad --aug=simclr \
     --synthetic --max_steps 200 --print_every 50 --evaluate_every 100000 --logdir /tmp/lr_sg$g 2>/dev/null | grep "img/s" | tail -2
done
