"""Training driver with the reference's CLI, experiment string and checkpoint layout
(/root/reference/main.py).  The hot loop (main.py:27-53 there) runs on libgmk kernels through
`model.train_step` (forward + fused sigmoid/MSE/KL loss kernel + backward, no autograd), a fused
AdamW launch and, under torchrun, a bucketed RCCL gradient all-reduce overlapped with backward.

    python main.py --latent_size 256 --sigmoid --wd 0.01 --epochs 100 --train --no_upsample --no_pool
    python main.py --latent_size 256 --sigmoid --wd 0.01 --epochs 100 --train --no_upsample --no_pool --variational
    torchrun --nproc_per_node 8 main.py ... (data parallel; every rank sees its own shuffled batches)

Kept quirks of the reference: checkpoint['epoch'] stores the completed epoch and resume starts at
it (main.py:221,264,328); beta grows per epoch (main.py:65-66); losses are printed x1000.
"""
import os
from datetime import datetime

import torch

from arguments import get_args_parser
from dataloaders import get_cifar
from metrics import compute_fid, compute_confidence_diversity, compute_inception_score
from models.autoencoders import Autoencoder
from utils import get_activation_function, plot_grid, make_folder
from gmk.optim import FusedAdamW
from gmk.parallel import GradReducer, init_from_env


def train_one_epoch(args, model, epoch=None, beta=None, train_dataloader=None, optim=None, lr_scheduler=None,
                    device=None, writer=None, reducer=None):
    model.train()
    total = torch.zeros(3, device=device)
    count = 0
    for image, label in train_dataloader:
        image = image.to(device)
        # forward + loss + backward (main.py:29-51): grads land in the optimizer's flat buffer
        losses = model.train_step(image, beta=beta if args.variational else 0.0, mse_prior=args.mse, reducer=reducer)
        if reducer is not None:
            reducer.wait()
        optim.step()
        lr_scheduler.step()
        total += losses
        if writer is not None:
            i = epoch * len(train_dataloader) + count
            rec, kl, tot = losses.tolist()
            if args.variational:
                writer.add_scalar('Train Loss/KL', kl, i)
            writer.add_scalar('Train Loss/Reconstruction', rec, i)
            writer.add_scalar('Train Loss/Total', tot, i)
            writer.add_scalar('Train Loss/beta', beta, i)
        count += 1
    if reducer is not None and reducer.world > 1:
        # every rank saw its own shard: report the mean over ranks (one 3-float all-reduce per epoch)
        from gmk.parallel import all_reduce_
        all_reduce_(total, group=reducer.group)
        total /= reducer.world
    avg_train_rec_loss = total[0] / len(train_dataloader)
    avg_train_kl_loss = beta * total[1] / len(train_dataloader) if args.variational else 0
    if args.variational:  # to combat vanishing KL loss
        beta *= args.beta_mult
    return beta, avg_train_rec_loss, avg_train_kl_loss


def compute_test_loss(args, model, beta=None, test_dataloader=None, device=None):
    """main.py:73-111: eval-mode losses plus mean |mu| and mean sigma of the latent.  Each batch is one no-grad
    forward + the fused loss kernel in its eval form (gmk_vae_loss_eval); the per-batch values are accumulated
    on the device, so the loop never synchronises."""
    model.eval()
    n = len(test_dataloader)
    acc = torch.zeros(5, device=device)   # rec, kl, rec + beta*kl, mean |mu|, mean sigma
    for test_image, _ in test_dataloader:
        acc += model.eval_losses(test_image.to(device), beta=beta if args.variational else 0.0, mse_prior=args.mse)
    rec, kl, _, latent_mean, latent_std = (acc / n).unbind()
    if not args.variational:
        kl = latent_mean = latent_std = 0
    return latent_mean, latent_std, rec, beta * kl


def make_checkpoint(model, optim, lr_scheduler, epoch, args, beta, loss_str):
    """The resume dict of /root/reference/main.py:323-331: same keys, same value types (torch.optim.AdamW /
    CosineAnnealingLR state_dict layouts, the completed epoch, the grown beta, the printed loss string)."""
    return dict(state_dict=model.state_dict(), optimizer=optim.state_dict(), lr_scheduler=lr_scheduler.state_dict(),
                epoch=epoch, args=args, current_beta=beta, losses=loss_str)


def compute_metrics(autoencoder=None, classifier=None, real_images=None, samples=None, debug=False):
    fid_ae = compute_fid(model=autoencoder, images1=real_images, images2=samples)
    fid_classifier = compute_fid(model=classifier, images1=real_images, images2=samples)
    confidence, diversity = compute_confidence_diversity(classifier, samples, debug=debug)
    inception_score = compute_inception_score(classifier=classifier, images=samples)
    return fid_classifier, fid_ae, confidence, diversity, inception_score


def main(args):
    for dir_name in ['checkpoints', 'data', 'plots', 'logs']:
        make_folder(dir_name)
    rank, local, world = init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit('this build runs on MI355X GPUs only (libgmk has no CPU path)')
    device = torch.device('cuda', local)

    # GPU-resident feeder: the dataset stays in HBM as uint8, batches are assembled (gather + crop + flip + /255) by one
    # kernel launch; with world > 1 every rank iterates its own shard of a shared epoch permutation
    train_dataloader, test_dataloader = get_cifar(data_dir=args.data_dir, train_batch_size=args.train_batch_size,
                                                  test_batch_size=args.test_batch_size, device=device, rank=rank,
                                                  world=world)
    if os.path.isfile('data/train_examples.pth') and os.path.isfile('data/test_examples.pth'):
        train_input_images = torch.load('data/train_examples.pth')
        test_input_images = torch.load('data/test_examples.pth')
    else:
        train_input_images = next(iter(train_dataloader))[0]
        test_input_images = next(iter(test_dataloader))[0]
        torch.save(train_input_images, 'data/train_examples.pth')
        torch.save(test_input_images, 'data/test_examples.pth')
    if test_input_images.shape[0] < args.latent_size:
        print(f'\n\nNumber of test images {test_input_images.shape[0]} should be >= number of '
              f'features {args.latent_size} in order to compute FID properly (only applies to VAE)\n\n')
        if args.variational:
            raise(SystemExit)

    args.act_fn = get_activation_function(act_str=args.act)
    if args.checkpoint is not None:
        print(f'\n\nLoading model checkpoint from {args.checkpoint}')
        # reference checkpoints pickle `args` (with an nn.Module inside): not weights_only-loadable
        checkpoint = torch.load(args.checkpoint, weights_only=False)
        # /root/reference/main.py:158-165: current against saved arguments, printed side by side
        saved_args = checkpoint['args']
        print(f'\n\nCurrent arguments:\n')
        for current_arg in vars(args):
            print(current_arg, getattr(args, current_arg))
        print(f'\n\nCheckpoint arguments:\n')
        for saved_arg in vars(saved_args):
            print(saved_arg, getattr(saved_args, saved_arg))

    model = Autoencoder(variational=args.variational, latent_size=args.latent_size, num_channels=args.num_channels,
                        kernel_size=args.kernel_size, act=args.act_fn, sigmoid=args.sigmoid,
                        interpolation=args.interpolation, no_pool=args.no_pool, no_upsample=args.no_upsample,
                        device=device, debug=args.debug).to(device)
    beta = args.beta
    init_epoch = 0
    if args.variational:
        model_type = f'vae_{args.beta}x{args.beta_mult}'
        if os.path.isfile(args.ae_checkpoint):
            print(f'\n\nLoading pretrained VAE for FID computation from {args.ae_checkpoint}')
            plain_ae = torch.load(args.ae_checkpoint, weights_only=False)
        else:
            print(f'\n\nTo compute FID score using a plain autoencoder, provide valid path to checkpoint, or train one:')
            print(f'\n\npython main.py --latent_size 256 --sigmoid --wd 0.01 --epochs 100 --train --no_upsample --no_pool\n\n')
            raise(SystemExit)
        from models.classifiers import evaluate_classifier, train_classifier
        classifier_checkpoint = args.classifier_checkpoint or f'checkpoints/cifar_classifier_{args.latent_size}.pth'
        if os.path.isfile(classifier_checkpoint):
            classifier = torch.load(classifier_checkpoint, weights_only=False)
            test_acc = evaluate_classifier(classifier, test_dataloader=test_dataloader, device=device)
            print(f'\n\nCIFAR-10 Test Accuracy {test_acc:.2f}')
        else:
            print(f'\n\nClassifier checkpoint is not found at {classifier_checkpoint}')
            classifier = train_classifier(args, classifier=None, train_dataloader=train_dataloader,
                                          test_dataloader=test_dataloader, device=device)
        if isinstance(plain_ae, dict) or isinstance(classifier, dict):
            raise NotImplementedError('\n\nmodel checkpoint must be a full saved model, not a state_dict\n\n')
    else:
        model_type = 'plain-ae'

    if args.checkpoint is not None:
        model.load_state_dict(checkpoint['state_dict'])
    if world > 1:
        for p in model.parameters():
            from gmk.parallel import broadcast_
            broadcast_(p.data, src=0)
    if args.train and getattr(args, "tune_kernels", False):
        # kernel-path choice for this device on a throw-away replica and synthetic input (gmk/tune.py): the real
        # parameters, the optimizer state and the data order are untouched
        from gmk.tune import tune_thin_paths
        # the replica's init and its eps draws go through torch's global generators: put them back afterwards, so that a tuned
        # run draws the same eps / shuffles as an untuned one
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(device)
        t_model = Autoencoder(variational=args.variational, latent_size=args.latent_size, num_channels=args.num_channels,
                              kernel_size=args.kernel_size, act=args.act_fn, sigmoid=args.sigmoid,
                              interpolation=args.interpolation, no_pool=args.no_pool, no_upsample=args.no_upsample,
                              device=device, debug=False).to(device)
        t_optim = FusedAdamW(t_model.parameters(), lr=args.lr, weight_decay=args.wd)
        t_image = torch.rand(args.train_batch_size, 3, 32, 32, device=device)

        def t_step():
            t_model.train_step(t_image, beta=beta if args.variational else 0.0, mse_prior=args.mse)
            t_optim.step()

        tuned = tune_thin_paths(t_step)
        if rank == 0:
            print(f'\n\nKernel paths on this device (thin_in, thin_outv): {tuned["choice"]}  {tuned["ms"]}')
        del t_model, t_optim, t_image, t_step
        torch.set_rng_state(rng_cpu)
        torch.cuda.set_rng_state(rng_dev, device)
    optim = FusedAdamW(model.parameters(), lr=args.lr, weight_decay=args.wd)
    lr_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=args.epochs * len(train_dataloader))
    reducer = GradReducer(model, optim) if world > 1 else None
    if args.checkpoint is not None:
        optim.load_state_dict(checkpoint['optimizer'])
        lr_scheduler.load_state_dict(checkpoint['lr_scheduler'])
        init_epoch = checkpoint['epoch']
        beta = checkpoint['current_beta']

    pool_str = 'pool-stride' if args.no_pool else 'pool-max'
    upsample_str = 'upsample-deconv' if args.no_upsample else f'upsample-{args.interpolation}'
    experiment_str = args.tag + f'{model_type}_latent{args.latent_size}_chan{args.num_channels}_{pool_str}_{upsample_str}_bs{args.train_batch_size}_lr{args.lr}_wd{args.wd}_e{args.epochs}'
    print(f'\n\n{experiment_str}')

    writer = None
    if args.log_tb:
        try:
            from torch.utils.tensorboard import SummaryWriter
            writer = SummaryWriter(args.tb_dir + '/' + experiment_str)
        except ImportError:
            print('tensorboard is not installed: --log_tb ignored')

    if args.evaluate:
        plot_grid(model, train_input_images[:4], name=experiment_str + '_train')
        plot_grid(model, test_input_images[:4], name=experiment_str + '_test')
    if args.sample:
        model.sample(name=experiment_str)

    if args.train:
        ref_metrics_str = ''
        if args.variational:
            fid_classifier, fid_ae, confidence, diversity, inception_score = compute_metrics(
                autoencoder=plain_ae, classifier=classifier, real_images=train_input_images.to(device),
                samples=test_input_images.to(device), debug=True)
            ref_metrics_str = f'Reference (real images) metrics:  FID (classifier/AE): ' + \
                f'{fid_classifier:.1f}/{fid_ae:.1f}  confidence {confidence:.1f}  diversity {diversity:.1f}  IS {inception_score:.3f}\n\n'
        print(f'\n\nTraining {model_type} model on CIFAR-10 images\n\n{ref_metrics_str}')

        for epoch in range(init_epoch, args.epochs, 1):
            beta, avg_train_rec_loss, avg_train_kl_loss = train_one_epoch(
                args, model, epoch=epoch, beta=beta, train_dataloader=train_dataloader, optim=optim,
                lr_scheduler=lr_scheduler, device=device, writer=writer, reducer=reducer)
            avg_latent_mean, avg_latent_std, avg_test_rec_loss, avg_test_kl_loss = compute_test_loss(
                args, model, beta=beta, test_dataloader=test_dataloader, device=device)
            metrics_str = ''
            if args.variational:
                samples = model.sample(num_images=args.test_batch_size, return_samples=True)
                fid_classifier, fid_ae, confidence, diversity, inception_score = compute_metrics(
                    autoencoder=plain_ae, classifier=classifier, real_images=test_input_images.to(device),
                    samples=samples, debug=True)
                metrics_str = f'  FID (classifier/AE): {fid_classifier:.1f}/{fid_ae:.1f}  confidence {confidence:.1f}  diversity {diversity:.1f}  IS {inception_score:.2f}'
            if writer is not None and rank == 0:   # reference main.py:299-311 (it raises NameError without --variational)
                i = epoch * len(train_dataloader)
                writer.add_scalar('Test Loss/KL', float(avg_test_kl_loss), i)
                writer.add_scalar('Test Loss/Reconstruction', float(avg_test_rec_loss), i)
                writer.add_scalar('Test Loss/Total', float(avg_test_rec_loss) + float(avg_test_kl_loss), i)
                writer.add_scalar('Test Loss/beta', beta, i)
                writer.add_scalar('Latent Vector/mean', float(avg_latent_mean), i)
                writer.add_scalar('Latent Vector/std', float(avg_latent_std), i)
                if args.variational:
                    writer.add_scalar('FID/classifier', float(fid_classifier), i)
                    writer.add_scalar('FID/plain_ae', float(fid_ae), i)
                    writer.add_scalar('Metrics/confidence', float(confidence), i)
                    writer.add_scalar('Metrics/diversity', float(diversity), i)
                    writer.add_scalar('Metrics/inception score', float(inception_score), i)
            latent_stats_str = f'  mean {avg_latent_mean:.4f} std {avg_latent_std:.4f}' if args.variational else ''
            kl_loss_str = f'  kl train {1000*avg_train_kl_loss:.2f} test {1000*avg_test_kl_loss:.2f}' if args.variational else ''
            loss_str = f'losses:  train {1000*avg_train_rec_loss:.2f} test {1000*avg_test_rec_loss:.2f}{kl_loss_str}'
            changes_str = f'  LR {lr_scheduler.get_last_lr()[0]:.5f}' + (f' beta {beta:.6f}' if args.variational else '')
            time_str = f'{str(datetime.now())[:-7]}'
            if rank == 0:
                print(f'{time_str}  Epoch {epoch:>3d}   {loss_str}{latent_stats_str}{metrics_str}{changes_str}')
                plot_grid(model, train_input_images[:4], name=experiment_str + '_train')
                plot_grid(model, test_input_images[:4], name=experiment_str + '_test')
                checkpoint = make_checkpoint(model, optim, lr_scheduler, epoch, args, beta, loss_str)
                torch.save(checkpoint, 'checkpoints/' + experiment_str + ".pth")
                model.sample(name=experiment_str)
        if rank == 0 and not os.path.isfile(args.ae_checkpoint):  # save full model checkpoint
            torch.save(model, 'checkpoints/' + experiment_str + "_full_model.pth")


if __name__ == "__main__":
    args = get_args_parser().parse_args()
    main(args)
