"""f-4 measurement (lives under tests/ because it times the oracle): the train transform on a 150 k-point scene, elastic
distortion forced on, device (segdino3d_amd.augment) vs the oracle (numpy + scipy, the reference's own arithmetic) on the host."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import augment as A
from segdino3d_amd.gtypes import GD3DTarget
from segdino3d_amd.synth import make_scene
from oracle import augment_ref as R

d = torch.device("cuda:0")
pts0, tgt0 = make_scene(2, 150_000, 3000, 300)
pts0[:, 3:] = (pts0[:, 3:] * 40 + 120).clamp(0, 255)
q0 = tgt0.extra_features["query2d_pos"].float()
tf = A.Compose3D([A.CustomRandomFlip3D(0.5, 0.5), A.CustomGlobalRotScaleTrans([-3.14, 3.14], [0.8, 1.2], [0.1, 0.1, 0.1]),
                  A.NormalizePointsColor(A.COLOR_MEAN, A.COLOR_STD), A.ElasticTransfrom([6, 20], [40, 160], 0.02, p=1.0), A.ToTensor()])


def run_dev(seed):
    np.random.seed(seed)
    p = pts0.to(d, non_blocking=True)
    t = GD3DTarget(extra_features={"query2d_pos": q0.to(d)})
    out, t = tf(p, t)
    return out, t


for s in range(3):
    run_dev(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
R_ = 10
for s in range(R_):
    out, t = run_dev(100 + s)
torch.cuda.synchronize()
ms_dev = 1e3 * (time.perf_counter() - t0) / R_
# kernels only (noise draw and upload excluded): affine + colour + voxel units + two displacements on resident data
p = pts0.to(d); el = A.ElasticTransfrom([6, 20], [40, 160], 0.02, p=1.0)
noise = [A.blurred_noise((70, 55, 30), device=d), A.blurred_noise((22, 18, 11), device=d)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    A.affine_(p, flip_x=True, angle=0.3, scale=1.05, trans=(0.1, 0.0, 0.0), color_mean=A.COLOR_MEAN, color_std=A.COLOR_STD)
    c = el._voxel_units(p)
    el._displace(c, noise[0], 6, 40); el._displace(c, noise[1], 20, 160)
e1.record(); torch.cuda.synchronize()
us_kernels = 1e3 * e0.elapsed_time(e1) / 20
np.random.seed(100)
t0 = time.perf_counter()
prm_forced = None
ref = R.draw_parameters()
x = pts0.numpy().copy(); x[:, :3] = R.affine(x[:, :3], ref); x[:, 3:] = R.normalize_color(x[:, 3:]); q = R.affine(q0.numpy(), ref)
coords = x[:, :3] / np.float32(0.02); qc = q / np.float32(0.02); np.random.rand()
for g, m in zip((6, 20), (40, 160)):
    nz, ax = R.elastic_noise(coords, g); coords = R.elastic_apply(coords, nz, ax, m); qc = R.elastic_apply(qc, nz, ax, m)
s_cpu = time.perf_counter() - t0
bytes_alg = 150_000 * (6 * 4 * 2 + 3 * 4 * 5)           # points read + written once, coordinates written once and read + written per pass
out = dict(points=150_000, ms_device_per_scene_incl_host_noise_and_upload=round(ms_dev, 2), us_kernels_only=round(us_kernels, 1),
           kernels_algorithmic_GBps=round(bytes_alg / us_kernels / 1e3, 1), s_cpu_oracle=round(s_cpu, 3), cpu_threads=1)
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/perf_augment.json", "w"), indent=1)
