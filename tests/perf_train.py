"""f-1 measurement (lives under tests/ because it times the oracle): the device criterion and the sparse-convolution
backward at training size, next to the oracle / a torch CPU model on the host cores.
  criterion: ScanNet200 training shape - 2250 queries (query_thr 0.5-1.0 of 3000 superpoints), 120 objects, 199 class
             logits, 7 prediction sets (initial + 6 decoder layers), sparse matcher with box costs;
  conv backward: the U-Net's layer shapes on a 150 k-point scene: forward, input gradient, weight gradient."""
import json, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops, train_ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
from oracle import loss_ref
from tests.loss_cases import as_pred
from tests.test_gpu_criterion import _training_size_case, build

d = torch.device("cuda:0")
out = {}


def timeit(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# ---------------------------------------------------------------- criterion
Q, S, G, n_cls, n_sem = 2250, 3000, 120, 198, 200
t, layers = _training_size_case(5, Q, S, G, n_cls, n_sem, n_layers=7)
cfg = dict(matcher="sparse", topk=1, cost_weights=[0.5, 1.0, 1.0, 0.5, 0.5], loss_weight=[0.5, 1.0, 1.0, 0.5, 0.5, 0.5], num_classes=n_cls,
           num_semantic_classes=n_sem, sem_ignore_index=n_sem, sem_loss_weight=0.5, non_object_weight=0.1, fix_dice_loss_weight=True,
           iter_matcher=True, fix_mean_loss=True)
crit = build(cfg)
t_d = {k: v.to(d) for k, v in t.items()}
l_d = [{k: [None if v is None else v.to(d).requires_grad_(True) for v in lst] for k, lst in layer.items()} for layer in layers]
pred = as_pred(l_d)


def step_dev():
    o = crit(pred, [t_d])
    (o["seg_loss"] + o["inst_loss"]).backward()


ms_dev = timeit(step_dev, 10)
l_c = [{k: [None if v is None else v.clone().requires_grad_(True) for v in lst] for k, lst in layer.items()} for layer in layers]
torch.set_num_threads(int(os.environ.get("SD3D_CPU_THREADS", "32")))
t0 = time.perf_counter()
o = loss_ref.unified_criterion(as_pred(l_c), [t], cfg)
(o["seg_loss"] + o["inst_loss"]).backward()
s_cpu = time.perf_counter() - t0
bytes_alg = 7 * (Q * S * 4 * 3)                     # logits read by the cost pass and the loss pass, gradient written
out["criterion"] = dict(ms_device=round(ms_dev, 3), s_cpu_oracle=round(s_cpu, 2), cpu_threads=torch.get_num_threads(),
                        shape=f"Q={Q} S={S} G={G} classes={n_cls + 1} layers=7", algorithmic_GBps=round(bytes_alg / ms_dev / 1e6, 1))

# ---------------------------------------------------------------- sparse convolution backward
pts, tgt = make_scene(0, 150000, 3000, 300)
maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
g = torch.Generator().manual_seed(0)
rows = []
for key, cin, cout in [(("same", 0, 3), 96, 96), (("same", 1, 3), 96, 96), (("same", 2, 3), 128, 128), (("same", 3, 3), 256, 256),
                       (("same", 4, 3), 256, 256), (("down", 0), 32, 32), (("up", 2), 256, 128)]:
    tab = maps.conv_table(*key)
    nbr, pairs = tab["nbr"], tab["pairs"]
    pairs_t = pairs if key[0] == "same" else maps.conv_table("up" if key[0] == "down" else "down", key[1])["pairs"]
    K, M = nbr.shape
    n_in = int(nbr.max().item()) + 1
    x = torch.randn(n_in, cin, generator=g).to(d)
    w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    dy = torch.randn(M, cout, generator=g).to(d)
    wt = train_ops.transposed_weights(w, key[0] == "same")
    P = int((pairs.in_idx >= 0).sum())
    fl = 2.0 * P * cin * cout
    us_f = 1e3 * timeit(lambda: ops.pair_conv(x, w, pairs), 5)
    us_dx = 1e3 * timeit(lambda: ops.pair_conv(dy, wt, pairs_t), 5)
    us_dw = 1e3 * timeit(lambda: train_ops.pair_wgrad(dy, x, pairs), 5)
    rows.append(dict(layer=f"{key} {cin}->{cout}", pairs=P, fwd_us=round(us_f), dgrad_us=round(us_dx), wgrad_us=round(us_dw),
                     fwd_TF=round(fl / us_f / 1e6, 1), dgrad_TF=round(fl / us_dx / 1e6, 1), wgrad_TF=round(fl / us_dw / 1e6, 1)))
out["sparse_conv_backward"] = rows
# ---------------------------------------------------------------- backbone training step (forward + backward)
import time as _t
import segdino3d_amd as seg
from segdino3d_amd.backbone_mink import Res16UNet34C
from oracle import sparse_ref as R
torch.manual_seed(0)
net = Res16UNet34C(in_channels=259, out_channels=96, config=dict(dilations=[1, 1, 1, 1], conv1_kernel_size=5, bn_momentum=0.02),
                   voxel_size=0.02, mode_fuse_2d_feat="early_fusion", add_positional_embedding=True).to(d).train()
rows = {}
for n_pts, n_sp in ((150000, 3000), (20000, 400)):
    pts_s, tgt_s = make_scene(1, n_pts, n_sp, 50)
    pd, td = pts_s.to(d), tgt_s.to(d)

    def step():
        for p_ in net.parameters():
            p_.grad = None
        f, _, _ = net.forward_wrapper([pd], [td], return_sp_mean_pos=True)
        (f[0] * f[0]).mean().backward()

    ms = timeit(step, 5)
    with seg.capture() as cap_:
        step()
    rows[f"{n_pts}_points"] = dict(ms_device_fwd_bwd=round(ms, 2), voxels=int(cap_.maps[0].n_vox[0]))
    if n_pts == 20000:                                   # the same step through the oracle + torch autograd on the host cores
        sd = {"backbone." + k: (v.detach().cpu().clone().requires_grad_(True) if v.is_floating_point() else v.cpu())
              for k, v in net.state_dict().items()}
        tgt_s = tgt_s.to("cpu")                          # Target.to moves in place
        R.BN_TRAIN = True
        t0 = _t.perf_counter()
        rf, _, _ = R.mink_forward_wrapper(sd, pts_s, tgt_s.extra_features["points_2dfeats"], tgt_s.extra_features["super_point_masks"])
        (rf * rf).mean().backward()
        rows[f"{n_pts}_points"]["s_cpu_oracle_fwd_bwd"] = round(_t.perf_counter() - t0, 2)
        rows[f"{n_pts}_points"]["cpu_threads"] = torch.get_num_threads()
        R.BN_TRAIN = False
out["res16unet34c_training_step"] = rows

# ---------------------------------------------------------------- whole training step (BASELINE configs[4] shape, one scene, fp32)
import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg
from segdino3d_amd.synth import add_training_targets
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(d).train()
steps = {}
for n_pts, n_sp, n_inst in ((150000, 3000, 40), (20000, 400, 10)):
    pts_s, tgt_s = make_scene(5, n_pts, n_sp, 300 if n_pts > 50000 else 50)
    tgt_s = add_training_targets(pts_s, tgt_s, n_instances=n_inst, seed=2)
    pd, td = pts_s.to(d), tgt_s.to(d)

    def train_step():
        for p_ in model.parameters():
            p_.grad = None
        for k_ in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
            td.__dict__.pop(k_, None)
        losses = model([pd], [td])
        (losses["seg_loss"] + losses["inst_loss"]).backward()
        return losses

    ms = timeit(train_step, 5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        for p_ in model.parameters():
            p_.grad = None
    fwd_only = []
    for _ in range(3):
        e0.record()
        with torch.no_grad():
            for k_ in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
                td.__dict__.pop(k_, None)
            model([pd], [td])
        e1.record(); torch.cuda.synchronize()
        fwd_only.append(e0.elapsed_time(e1))
    with seg.capture() as cap_:
        l = train_step()
    steps[f"{n_pts}_points"] = dict(ms_forward_loss_backward=round(ms, 2), ms_forward_loss_no_grad=round(min(fwd_only), 2),
                                    queries=int(cap_.outputs["masks"][0].shape[0]), objects=int(td.labels.shape[0]),
                                    seg_loss=round(float(l["seg_loss"]), 4), inst_loss=round(float(l["inst_loss"]), 4))
out["full_model_training_step"] = steps
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/perf_train.json", "w"), indent=1)
