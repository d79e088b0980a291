"""Command-line options — Py3 counterpart of /root/reference/augmented_cyclegan/options.py (which is Python 2:
`import cPickle`, options.py:4).  Same flags, defaults, `opt.txt` format, `opt.pkl`, sub-directory creation
(options.py:7-12, 20-131).  Additions (not in the reference): --n_blocks, --precision, --synthetic, --dist."""
import argparse
import os
import pickle

import torch


def create_sub_dirs(opt, sub_dirs):
    """options.py:7-12"""
    for sub_dir in sub_dirs:
        dir_path = os.path.join(opt.expr_dir, sub_dir)
        os.makedirs(dir_path, exist_ok=True)
        setattr(opt, sub_dir, dir_path)


# flag table: (name, kind, default, help).  kind is a type, a ("choice", type, values) tuple, or "flag".
# Values and order follow options.py:20-85 of the reference; the last group is new here.
_T = [
    ("dataroot", str, None, "path to data (trainA/B.npz, testA/B.npz)"),
    ("checkpoints_dir", str, "./checkpoints/", "models are saved here"),
    # data
    ("input_nc", int, 3, "# of input image channels"),
    ("output_nc", int, 3, "# of output image channels"),
    ("grid_size", int, 256, "resolution of input/output grids"),
    ("numpy_data", ("choice", int, [0, 1]), 1, "use numpy data"),
    # experiment
    ("seed", int, None, "manual seed"),
    ("model", ("choice", str, ["cycle_gan", "stoch_cycle_gan", "aug_cycle_gan"]), "aug_cycle_gan", "which model to train"),
    ("gpu_ids", str, "0", 'gpu ids: e.g. 0  0,1,2 (only "on the GPU" matters here; -1 is rejected later: no CPU path)'),
    # supervised training
    ("supervised", "flag", False, "also run the paired step"),
    ("sup_frac", float, 0.1, "fraction of training data for supervised training"),
    ("lambda_sup_A", float, 0.1, "weight for supervised loss (B -> A)"),
    ("lambda_sup_B", float, 0.1, "weight for supervised loss (A -> B)"),
    # training
    ("batchSize", int, 32, "input batch size"),
    ("continue_train", "flag", False, "reload <expr_dir>/<which_epoch> before training"),
    ("which_epoch", str, "latest", "checkpoint name to resume from"),
    ("epoch_count", int, 1, "the starting epoch count"),
    ("niter", int, 25, "# of epochs at starting learning rate"),
    ("niter_decay", int, 25, "# of epochs to linearly decay learning rate to zero"),
    ("beta1", float, 0.5, "momentum term of adam"),
    ("lr", float, 0.0002, "initial learning rate for adam"),
    # model
    ("ngf", int, 32, "# of gen filters in first conv layer"),
    ("nef", int, 32, "# of encoder filters in first conv layer"),
    ("ndf", int, 64, "# of discrim filters in first conv layer"),
    ("nlatent", int, 16, "# of latent code dimensions"),
    ("which_model_netD", str, "basic", "(accepted, unused — as in the reference)"),
    ("which_model_netG", str, "resnet", "(accepted, unused — as in the reference)"),
    ("norm", str, "instance", "instance or batch normalization"),
    ("use_dropout", "flag", False, "use dropout for the generator"),
    ("max_gnorm", float, 500., "max grad norm to which it will be clipped"),
    ("stoch_enc", "flag", False, "use a stochastic encoder"),
    ("z_gan", ("choice", int, [0, 1]), 1, "use a GAN on z_B"),
    ("enc_A_B", ("choice", int, [0, 1]), 1, "encoder of z_B conditioned on both A and B"),
    ("no_lsgan", "flag", False, "vanilla GAN (the reference's BCE branch is broken; raises here)"),
    ("lambda_A", float, 1.0, "weight for cycle loss (A -> B -> A)"),
    ("lambda_B", float, 1.0, "weight for cycle loss (B -> A -> B)"),
    ("lambda_z_B", float, 0.025, "weight for the latent cycle loss"),
    # monitoring
    ("monitor_gnorm", bool, True, "monitor grad norms (type=bool as in options.py:77: any string is True)"),
    ("display_freq", int, 5000, "frequency of PNG dumps"),
    ("print_freq", int, 100, "frequency of log lines"),
    ("save_epoch_freq", int, 5, "frequency of saving checkpoints at the end of epochs"),
    ("num_multi", int, 10, "the number of z_B used to generate different B"),
    ("eval_A_freq", int, 1, "frequency of evaluating on A"),
    ("eval_B_freq", int, 1, "frequency of evaluating on B"),
    # additions of this implementation
    ("n_blocks", int, 3, "residual blocks per generator (the reference builds 3)"),
    ("precision", ("choice", str, ["bf16x3", "f32"]), "bf16x3",
     "conv arithmetic on the matrix cores: bf16x3 (default; split-bf16 products, inside the 1e-3 parity bar — what bench.py "
     "and the parity tests run), f32 (exact products, 2.4x slower)"),
    ("synthetic", int, 0, "use N synthetic U(-1,1) samples per split instead of --dataroot"),
    ("sync_bn", "flag", False, "data parallel: BatchNorm (E_B, D_z_B) statistics across all ranks"),
    ("step_graph", "flag", False, "replay the training step as one captured HIP graph (launch-bound sizes: small images / "
                                  "batches; single GPU)"),
    ("eval_steps", int, 50, "variational-bound steps per epoch (train.py:285 uses 50)"),
]


class TrainOptions(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.initialized = False

    def initialize(self):
        self.parser.add_argument("--name", type=str, required=True, help="name of the experiment")
        for name, kind, default, text in _T:
            if kind == "flag":
                self.parser.add_argument("--" + name, action="store_true", help=text)
            elif isinstance(kind, tuple):
                self.parser.add_argument("--" + name, type=kind[1], choices=kind[2], default=default, help=text)
            else:
                self.parser.add_argument("--" + name, type=kind, default=default, help=text)
        self.initialized = True

    def parse(self, sub_dirs=None, argv=None):
        if not self.initialized:
            self.initialize()
        opt = self.opt = self.parser.parse_args(argv)
        if opt.dataroot is None and not opt.synthetic:
            self.parser.error("--dataroot is required (or --synthetic N)")
        opt.gpu_ids = [i for i in (int(tok) for tok in opt.gpu_ids.split(",")) if i >= 0]      # options.py:92-97
        if opt.gpu_ids and torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", opt.gpu_ids[0]))
            torch.cuda.set_device(local)
            opt.gpu_ids = [local]
        opt.expr_dir = os.path.join(opt.checkpoints_dir, opt.name)
        os.makedirs(opt.expr_dir, exist_ok=True)
        args = vars(opt)
        banner = ["------------ Options -------------"]
        banner += ["%s: %s" % (str(k), str(args[k])) for k in sorted(args)]
        banner += ["-------------- End ----------------"]
        print("\n".join(banner))
        with open(os.path.join(opt.expr_dir, "opt.txt"), "wt") as f:                          # options.py:119-124
            f.write("\n".join(banner) + "\n")
        with open(os.path.join(opt.expr_dir, "opt.pkl"), "wb") as f:                          # options.py:126-128
            pickle.dump(args, f)
        if sub_dirs is not None:
            create_sub_dirs(opt, sub_dirs)
        return opt


class TestOptions(object):
    """options.py:134-143"""

    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.parser.add_argument('--chk_path', required=True, type=str)
        self.parser.add_argument('--res_dir', type=str, default='test_res')
        self.parser.add_argument('--train_logvar', type=int, default=1)
        self.parser.add_argument('--dataroot', required=True, type=str)
        self.parser.add_argument('--metric', required=True, type=str, choices=['bpp', 'mse', 'visual', 'noise_sens'])

    def parse(self, argv=None):
        return self.parser.parse_args(argv)
