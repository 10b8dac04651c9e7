"""Debug helper: configs[0] instance-count difference between the HIP path and the oracle (where do the rows drop?)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import segdino3d_amd as seg
from oracle import model_ref, decoder_ref as D
from segdino3d_amd.configs import scannetv2_model_cfg
from segdino3d_amd.synth import make_scene, structure_scene, sharpen_random_model
d = torch.device("cuda:0")
pts, tgt = make_scene(21, 10000, 300, 50)
structure_scene(pts, tgt)
torch.manual_seed(0)
model = sharpen_random_model(seg.build_architecture(scannetv2_model_cfg(query_num=-1)).eval())
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
model.to(d); model.to_host = False
with torch.no_grad(), seg.capture() as cap:
    pd = model([pts.to(d)], [tgt.to(d)])[0].pred_pts_seg
tgt = tgt.to("cpu"); ef = tgt.extra_features
ref, mid = model_ref.forward_eval(sd, pts, ef["points_2dfeats"], ef["super_point_masks"], ef["query2d_feats"], ef["query2d_pos"], tgt.masks,
                                  backbone="spconv", query_num=-1, num_classes=18, dec_cfg=D.DecoderCfg(normalize_box_prediction=False), return_intermediate=True)
q_h, sm_h, nm_h = pd.sort_and_mask
q_r, sm_r, nm_r = ref["sort_and_mask"]
print("score_mask true: hip", int(sm_h.sum()), "ref", int(sm_r.sum()), "| npoint_mask true: hip", int(nm_h.sum()), "ref", int(nm_r.sum()), "of", nm_h.numel(), nm_r.numel())
hs = np.sort(pd.instance_scores.cpu().numpy()); rs = np.sort(ref["instance_scores"].numpy())
print("smallest scores hip", hs[:8], "ref", rs[:20])
print("ref scores < 1e-30:", int((rs < 1e-30).sum()), " hip:", int((hs < 1e-30).sum()))
cls_h, cls_r = cap.outputs["cls_preds"][0].cpu(), mid["decoder"]["cls_preds"]
print("cls logits range", float(cls_r.min()), float(cls_r.max()))
