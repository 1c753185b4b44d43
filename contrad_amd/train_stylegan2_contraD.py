"""``train_stylegan2_contraD.py`` of the reference (StyleGAN2 + ContraD with the fused G_D call structure) on the
MI355X path -- see contrad_amd/train_stylegan2.py, which holds both loops."""
from .train_stylegan2 import main as _main


def main(argv=None):
    return _main(argv, contrad_script=True)


if __name__ == '__main__':
    main()
