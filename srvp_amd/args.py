"""
Command-line surface of the training program: the 47 flags of the reference's args.py:42-163 with the same names,
types, defaults and required-ness (plain argparse; the AMP / Apex flags still parse but are no-ops because the HIP
kernels are natively bf16-operand / fp32-accumulate).  Additions: `--local-rank` (torch >= 2 launcher spelling) and the
`synthetic` dataset used by bench.py / smoke tests (the reference's data loaders are out of scope).
"""
import argparse

ARCH_TYPES = ['dcgan', 'vgg']
DATASETS = ['smmnist', 'kth', 'human', 'bair', 'synthetic']
AMP_OPT_LEVELS = ['O0', 'O1', 'O2', 'O3']

# (flag, kwargs) in the reference's order
_FLAGS = [
    ('--seed', dict(type=int, default=None, help='manual seed (random if omitted)')),
    ('--save_path', dict(type=str, required=True, help='directory receiving model.pt / model_best.pt / model_<itr>.pt')),
    ('--torch_amp', dict(action='store_true', help='accepted for compatibility; no-op')),
    ('--apex_amp', dict(action='store_true', help='accepted for compatibility; no-op')),
    ('--amp_opt_lvl', dict(type=str, default='O1', choices=AMP_OPT_LEVELS, help='accepted for compatibility; no-op')),
    ('--keep_batchnorm_fp32', dict(action='store_true', default=None, help='accepted for compatibility; no-op')),
    ('--apex_verbose', dict(action='store_true', help='accepted for compatibility; no-op')),
    ('--local_rank', dict(type=int, default=0, help='process rank on this node (also read from LOCAL_RANK)')),
    ('--device', dict(type=int, default=None, nargs='+', help='GPU indices, one process per entry')),
    ('--n_workers', dict(type=int, default=4, help='data-loader worker processes')),
    ('--nhx', dict(type=int, default=128, help='size of the frame encodings')),
    ('--ny', dict(type=int, required=True, help='size of the state variable y')),
    ('--nz', dict(type=int, required=True, help='size of the auxiliary random variable z')),
    ('--n_euler_steps', dict(type=int, default=1, help='Euler sub-steps per frame')),
    ('--nt_inf', dict(type=int, required=True, help='frames used to infer y_1 and the content variable')),
    ('--obs_scale', dict(type=float, default=1, help='standard deviation of the observation model')),
    ('--archi', dict(type=str, default='dcgan', choices=ARCH_TYPES, help='encoder / decoder family')),
    ('--skipco', dict(action='store_true', help='skip connections from encoder to decoder')),
    ('--nf', dict(type=int, default=64, help='base filter count')),
    ('--nh_res', dict(type=int, default=512, help='hidden size of the residual MLP f')),
    ('--nlayers_res', dict(type=int, default=4, help='layers of the residual MLP f')),
    ('--nh_inf', dict(type=int, default=256, help='hidden size of the inference networks')),
    ('--nlayers_inf', dict(type=int, default=3, help='layers of the inference MLPs')),
    ('--res_gain', dict(type=float, default=1.41, help='orthogonal-init gain of the residual MLP')),
    ('--beta_y', dict(type=float, default=1, help='weight of KL(q(y_1) || N(0, I))')),
    ('--beta_z', dict(type=float, default=1, help='weight of KL(q(z) || p(z))')),
    ('--l2_res', dict(type=float, default=1, help='weight of the residual L2 penalty')),
    ('--batch_size', dict(type=int, default=128, help='global training batch size')),
    ('--lr', dict(type=float, default=0.0003, help='Adam learning rate')),
    ('--lr_scheduling_burnin', dict(type=int, default=1000000, help='steps before the learning rate decays')),
    ('--lr_scheduling_n_iter', dict(type=int, default=100000, help='steps of linear decay to zero')),
    ('--dataset', dict(type=str, required=True, choices=DATASETS, help='dataset name')),
    ('--data_dir', dict(type=str, required=True, help='dataset directory')),
    ('--seq_len', dict(type=int, required=True, help='training sequence length')),
    ('--ndigits', dict(type=int, default=2, help='Moving MNIST: number of digits')),
    ('--max_speed', dict(type=int, default=4, help='Moving MNIST: maximum digit speed')),
    ('--deterministic', dict(action='store_true', help='Moving MNIST: deterministic bounces')),
    ('--subsampling', dict(type=int, default=8, help='Human3.6M: temporal subsampling')),
    ('--nx', dict(type=int, default=64, help='frame size')),
    ('--nc', dict(type=int, required=True, help='image channels')),
    ('--val_interval', dict(type=int, default=20000, help='steps between validations')),
    ('--chkpt_interval', dict(type=int, default=None, help='steps between intermediate checkpoints')),
    ('--batch_size_test', dict(type=int, default=16, help='validation batch size')),
    ('--n_iter_test', dict(type=int, default=25, help='validation batches per evaluation')),
    ('--nt_cond', dict(type=int, required=True, help='conditioning frames at test time')),
    ('--n_samples_test', dict(type=int, default=100, help='predictions per video during validation')),
    ('--seq_len_test', dict(type=int, default=None, help='validation sequence length (default: seq_len)')),
]


def create_args():
    p = argparse.ArgumentParser(prog='srvp_amd.train', description='SRVP training on MI355X (HIP kernels)',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flag, kw in _FLAGS:
        if flag == '--local_rank':
            p.add_argument('--local_rank', '--local-rank', dest='local_rank', **kw)
        else:
            p.add_argument(flag, **kw)
    return p
