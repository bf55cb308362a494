"""End-to-end demo on the GPU box: a fake MNIST file (procedural 'digits'), then the reference CLI (python -m srvp_amd.train)
on the Stochastic Moving-MNIST generator with the reference's SM-MNIST recipe (README.md:111-119) for a few hundred iterations;
prints the training log (loss, validation -PSNR).  usage: python tools/train_smmnist_demo.py [n_iter]"""
import gzip, os, struct, subprocess, sys, tempfile
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 600
d = tempfile.mkdtemp()
rng = np.random.RandomState(0)
imgs = np.zeros((2000, 28, 28), np.uint8)
yy, xx = np.mgrid[0:28, 0:28]
for i in range(2000):
    k = rng.randint(3)
    cy, cx, r = rng.uniform(10, 18), rng.uniform(10, 18), rng.uniform(5, 9)
    if k == 0:
        m = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    elif k == 1:
        m = (abs(yy - cy) < r) & (abs(xx - cx) < r * 0.5)
    else:
        m = (abs(yy - cy) + abs(xx - cx) < r) | ((abs(yy - cy) < 1.5) & (abs(xx - cx) < r))
    imgs[i][m] = rng.randint(160, 256)
os.makedirs(os.path.join(d, 'MNIST', 'raw'))
with gzip.open(os.path.join(d, 'MNIST', 'raw', 'train-images-idx3-ubyte.gz'), 'wb') as f:
    f.write(struct.pack('>iiii', 2051, 2000, 28, 28) + imgs.tobytes())
cmd = [sys.executable, '-m', 'srvp_amd.train', '--device', '0', '--seed', '1', '--dataset', 'smmnist', '--data_dir', d, '--save_path', os.path.join(d, 'run'),
       '--nc', '1', '--seq_len', '15', '--nt_cond', '5', '--nt_inf', '5', '--ny', '20', '--nz', '20', '--archi', 'dcgan', '--beta_z', '2',
       '--batch_size', '128', '--batch_size_test', '16', '--n_iter_test', '4', '--n_samples_test', '10', '--seq_len_test', '25',
       '--lr_scheduling_burnin', str(n_iter), '--lr_scheduling_n_iter', '1', '--val_interval', str(max(1, n_iter // 4)), '--n_euler_steps', '1']
print(' '.join(cmd), flush=True)
sys.exit(subprocess.call(cmd, cwd=root))
