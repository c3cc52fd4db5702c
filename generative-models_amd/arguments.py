"""Command-line flags: names, defaults and semantics of /root/reference/arguments.py:7-38."""
import argparse


def get_args_parser(add_help=True):
    parser = argparse.ArgumentParser(description="PyTorch Detection Training", add_help=add_help)
    a = parser.add_argument
    a("--data_dir", default="data", type=str, help="path to dataset")
    a("--checkpoint", default=None, type=str, help="path to model checkpoint")
    a("--ae_checkpoint", default='checkpoints/plain-ae_latent256_chan64_pool-stride_upsample-deconv_bs50_lr0.001_wd0.01_e100_full_model.pth', type=str, help="path to AE checkpoint")
    a("--classifier_checkpoint", default=None, type=str, help="path to classifier checkpoint")
    a("--tag", default="", type=str, help="string to prepend when saving checkpoints")
    a("--debug", dest="debug", help="print out shapes and values of intermediate outputs", action="store_true")
    a("--log_tb", dest="log_tb", help="enable tensorboard plotting", action="store_true")
    a("--tb_dir", default="logs", type=str, help="path to tensorboard files")
    a("--loss", default="mse", type=str, help="reconstruction loss function")
    a("--train_batch_size", default=50, type=int)
    a("--test_batch_size", default=500, type=int)
    a("--latent_size", default=256, type=int)
    a("--num_channels", default=64, type=int)
    a("--kernel_size", default=3, type=int)
    a("--print_freq", default=1000, type=int)
    a("--epochs", default=100, type=int)
    a("--lr", default=0.001, type=float)
    a("--wd", default=0.0, type=float)
    a("--dropout", default=0, type=float)
    a("--act", default='gelu', type=str, help='relu, leaky-relu, elu, selu, gelu, swish, mish')
    a("--beta", default=5e-6, type=float)
    a("--beta_mult", default=1.02, type=float)
    a("--sample", dest="sample", help="generate an image from a random latent vector", action="store_true")
    a("--evaluate", dest="evaluate", help="use pretrained model to reconstruct 4 images", action="store_true")
    a("--train", dest="train", help="train model", action="store_true")
    a("--variational", dest="variational", help="use variational autoencoder, otherwise use a plain one", action="store_true")
    a("--sigmoid", dest="sigmoid", help="apply sigmoid at the end of decoder", action="store_true")
    a("--mse", dest="mse", help="ise MSE loss instead of KL loss", action="store_true")
    a("--no_pool", dest="no_pool", help="don't use max pooling for downsampling, use stride=2 conv", action="store_true")
    a("--no_upsample", dest="no_upsample", help="don't use nn.Upsample for upsampling, use deconv", action="store_true")
    a("--interpolation", default="nearest", type=str, help="upsample interpolation mode in the decoder, "
      "'nearest', 'linear', 'bilinear', 'bicubic' and 'trilinear'")
    # not in the reference: time the train step with both kernel paths of the 3-channel-side layers on this device
    # and keep the faster one (gmk/tune.py); off by default so that runs stay bitwise comparable across devices
    a("--tune_kernels", dest="tune_kernels", help="per-device kernel-path tuning before training", action="store_true")
    return parser
