"""Device criterion (segdino3d_amd/criterion.py + csrc/loss.hip) against the reference's own
outputs (tests/golden/loss_criterion.npz: losses and autograd gradients of loss_3d.py) and, at training size,
against the float64 oracle.  Tolerances: fp32 sums of ~S terms -> 2e-5 relative on losses, 2e-5 of the largest
gradient entry on gradients; matches (integer decisions) must be identical."""
import pytest
import torch

from tests.loss_cases import KEYS, as_pred, load_case

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def build(cfg):
    from segdino3d_amd.criterion import ScanNetUnifiedCriterion
    costs = ["QueryClassificationCost", "MaskBCECost", "MaskDiceCost", "CenterL1Cost", "SizeL1Cost"]
    matcher = dict(type="SparseMatcher" if cfg["matcher"] == "sparse" else "HungarianMatcher",
                   costs=[dict(type=t, weight=w) for t, w in zip(costs, cfg["cost_weights"])])
    if cfg["matcher"] == "sparse":
        matcher["topk"] = cfg["topk"]
    return ScanNetUnifiedCriterion(
        num_semantic_classes=cfg["num_semantic_classes"],
        sem_criterion=dict(type="ScanNetSemanticCriterion", ignore_index=cfg["sem_ignore_index"], loss_weight=cfg["sem_loss_weight"]),
        inst_criterion=dict(type="InstanceCriterion", matcher=matcher, loss_weight=cfg["loss_weight"], num_classes=cfg["num_classes"],
                            non_object_weight=cfg["non_object_weight"], fix_dice_loss_weight=cfg["fix_dice_loss_weight"],
                            iter_matcher=cfg["iter_matcher"], fix_mean_loss=cfg["fix_mean_loss"]))


@pytest.mark.parametrize("name", ["s200", "base", "hung"])
def test_criterion_matches_reference_outputs(name):
    d = dev()
    cfg, targets, layers, exp = load_case(name, torch.float32, d)
    for layer in layers:
        for k in KEYS:
            layer[k] = [None if v is None else v.requires_grad_(True) for v in layer[k]]
    crit = build(cfg)
    out = crit(as_pred(layers), targets)
    assert abs(float(out["seg_loss"].detach()) - exp["seg_loss"]) < 2e-5 * abs(exp["seg_loss"])
    assert abs(float(out["inst_loss"].detach()) - exp["inst_loss"]) < 2e-5 * abs(exp["inst_loss"])
    (out["seg_loss"] + out["inst_loss"]).backward()
    for l, layer in enumerate(layers):
        for k in KEYS:
            for b, v in enumerate(layer[k]):
                if v is None:
                    continue
                ref = exp["grads"][l][k][b]
                g = v.grad.cpu() if v.grad is not None else torch.zeros_like(ref)
                assert (g - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1e-3), (l, k, b)


def test_matches_equal_the_oracles():
    from oracle import loss_ref
    d = dev()
    for name in ("s200", "base"):
        cfg, targets, layers, _ = load_case(name, torch.float32, d)
        crit = build(cfg)
        with torch.no_grad():
            crit(as_pred(layers), targets)
        cfg64, t64, l64, _ = load_case(name, torch.float64)
        ref = loss_ref.unified_criterion(as_pred(l64), t64, cfg64)
        for b, (iq, ig) in enumerate(ref["_indices"]):
            m = crit.last_matches[0][b].cpu()
            exp = torch.zeros_like(m)
            exp[iq, ig] = 1
            assert torch.equal(m, exp), (name, b)


def _training_size_case(seed, Q, S, G, n_cls, n_sem, n_layers=2):
    g = torch.Generator().manual_seed(seed)
    owner = torch.randint(0, G + 1, (S,), generator=g)
    inst = torch.stack([owner == k for k in range(G)])
    sem_id = torch.randint(0, n_sem + 1, (S,), generator=g)
    sem = torch.stack([sem_id == k for k in range(n_sem + 1)])
    sp = torch.cat([inst, sem])
    ids = torch.randperm(S, generator=g)[:Q]
    t = dict(sp_inst_sem_masks=sp, query_inst_sem_masks=sp[:, ids], labels=torch.randint(0, n_cls, (G,), generator=g),
             instance_centers=torch.rand(G, 3, generator=g) * 6, instance_sizes=torch.rand(G, 3, generator=g) * 2)
    same = (inst.float().T @ inst[:, ids].float()).T                            # [Q, S]
    layers = []
    for l in range(n_layers):
        layers.append(dict(cls_preds=[torch.randn(Q, n_cls + 1, generator=g)], sem_preds=[torch.randn(Q, n_sem + 1, generator=g)],
                           masks=[(same * 2 - 1) * 1.5 + 2.0 * torch.randn(Q, S, generator=g)], scores=[torch.rand(Q, 1, generator=g)],
                           centers=[torch.rand(Q, 3, generator=g) * 6 if l else None], sizes=[torch.rand(Q, 3, generator=g) * 2 if l else None]))
    return t, layers


def test_training_size_against_float64_oracle():
    """ScanNet200 training shape: ~2250 queries (query_thr 0.5-1.0 of 3000 superpoints), 120 objects, 198 + 1 classes."""
    from oracle import loss_ref
    d = dev()
    Q, S, G, n_cls, n_sem = 2250, 3000, 120, 198, 200
    t, layers = _training_size_case(5, Q, S, G, n_cls, n_sem)
    cfg = dict(matcher="sparse", topk=1, cost_weights=[0.5, 1.0, 1.0, 0.5, 0.5], loss_weight=[0.5, 1.0, 1.0, 0.5, 0.5, 0.5], num_classes=n_cls,
               num_semantic_classes=n_sem, sem_ignore_index=n_sem, sem_loss_weight=0.5, non_object_weight=0.1, fix_dice_loss_weight=True,
               iter_matcher=True, fix_mean_loss=True)
    crit = build(cfg)
    t_d = {k: v.to(d) for k, v in t.items()}
    l_d = [{k: [None if v is None else v.to(d).requires_grad_(True) for v in lst] for k, lst in layer.items()} for layer in layers]
    out = crit(as_pred(l_d), [t_d])
    (out["seg_loss"] + out["inst_loss"]).backward()
    t64 = {k: (v.double() if v.is_floating_point() else v) for k, v in t.items()}
    l64 = [{k: [None if v is None else v.double().requires_grad_(True) for v in lst] for k, lst in layer.items()} for layer in layers]
    ref = loss_ref.unified_criterion(as_pred(l64), [t64], cfg)
    (ref["seg_loss"] + ref["inst_loss"]).backward()
    assert abs(float(out["seg_loss"].detach()) - float(ref["seg_loss"].detach())) < 2e-5 * abs(float(ref["seg_loss"].detach()))
    assert abs(float(out["inst_loss"].detach()) - float(ref["inst_loss"].detach())) < 2e-5 * abs(float(ref["inst_loss"].detach()))
    for a, b in zip(l_d, l64):
        for k in KEYS:
            if a[k][0] is None:
                continue
            gr = b[k][0].grad if b[k][0].grad is not None else torch.zeros_like(b[k][0])
            ga = a[k][0].grad.cpu().double() if a[k][0].grad is not None else torch.zeros_like(gr)
            assert (ga - gr).abs().max().item() <= 2e-5 * max(gr.abs().max().item(), 1e-6), k
    # the criterion is deterministic: a second evaluation returns the same bits
    out2 = crit(as_pred(l_d), [t_d])
    assert torch.equal(out["inst_loss"].detach(), out2["inst_loss"].detach())


def test_cpu_tensors_are_refused():
    dev()
    cfg, targets, layers, _ = load_case("base", torch.float32, "cpu")
    crit = build(cfg)
    with pytest.raises(RuntimeError):
        crit(as_pred(layers), targets)


def _tiny_case(G, Q=24, S=40, n_cls=5, n_sem=6, seed=0, empty_object=False):
    g = torch.Generator().manual_seed(seed)
    owner = torch.randint(0, max(G, 1), (S,), generator=g) if G else torch.zeros(S, dtype=torch.long)
    inst = torch.stack([owner == k for k in range(G)]) if G else torch.zeros(0, S, dtype=torch.bool)
    sem_id = torch.randint(0, n_sem + 1, (S,), generator=g)
    sem = torch.stack([sem_id == k for k in range(n_sem + 1)])
    sp = torch.cat([inst, sem])
    ids = torch.randperm(S, generator=g)[:Q]
    qm = sp[:, ids].clone()
    if empty_object and G:
        qm[0] = False                                               # no query lies inside object 0
    t = dict(sp_inst_sem_masks=sp, query_inst_sem_masks=qm, labels=torch.randint(0, n_cls, (G,), generator=g))
    layer = dict(cls_preds=[torch.randn(Q, n_cls + 1, generator=g)], sem_preds=[torch.randn(Q, n_sem + 1, generator=g)],
                 masks=[torch.randn(Q, S, generator=g)], scores=[None], centers=[None], sizes=[None])
    cfg = dict(matcher="sparse", topk=1, cost_weights=[0.5, 1.0, 1.0], loss_weight=[0.5, 1.0, 1.0, 0.5], num_classes=n_cls,
               num_semantic_classes=n_sem, sem_ignore_index=n_sem, sem_loss_weight=0.5, non_object_weight=0.1, fix_dice_loss_weight=True,
               iter_matcher=True, fix_mean_loss=True)
    return t, layer, cfg


def test_scene_without_objects_behaves_like_the_reference():
    """No ground-truth object: the reference's mask terms are means over nothing (NaN, loss_3d.py:479-481 on empty tensors) while
    the class term is finite (every query -> "no object"); the device criterion returns the same and finite class gradients."""
    from oracle import loss_ref
    d = dev()
    t, layer, cfg = _tiny_case(G=0)
    crit = build(cfg)
    l_d = {k: [None if v is None else v.to(d).requires_grad_(True) for v in lst] for k, lst in layer.items()}
    out = crit(dict(l_d), [{k: v.to(d) for k, v in t.items()}])
    ref = loss_ref.unified_criterion(dict(layer), [t], cfg)
    assert torch.isnan(out["inst_loss"]) and torch.isnan(ref["inst_loss"])
    assert abs(float(out["seg_loss"].detach()) - float(ref["seg_loss"])) < 1e-5
    parts = crit.last_parts[0].cpu()
    assert abs(float(parts[0]) - float(ref["_parts"][0][0])) < 1e-5 and bool(torch.isnan(parts[1]))


def test_object_without_any_query_inside_is_left_unmatched():
    """`query_masks` row all False -> every cost of that object is 1e8 -> `cost < kth` selects nothing (loss_3d.py:358-364)."""
    from oracle import loss_ref
    d = dev()
    t, layer, cfg = _tiny_case(G=4, empty_object=True, seed=3)
    crit = build(cfg)
    l_d = {k: [None if v is None else v.to(d).requires_grad_(True) for v in lst] for k, lst in layer.items()}
    out = crit(dict(l_d), [{k: v.to(d) for k, v in t.items()}])
    m = crit.last_matches[0][0].cpu()
    assert int(m[:, 0].sum()) == 0 and int(m.sum()) == 3
    t64 = {k: v for k, v in t.items()}
    l64 = {k: [None if v is None else v.double() for v in lst] for k, lst in layer.items()}
    ref = loss_ref.unified_criterion(dict(l64), [t64], cfg)
    assert abs(float(out["inst_loss"].detach()) - float(ref["inst_loss"])) < 2e-5 * abs(float(ref["inst_loss"]))


def test_shared_matches_when_iter_matcher_is_off():
    """iter_matcher=False: the auxiliary layers reuse the last layer's matches (loss_3d.py:705-708)."""
    from oracle import loss_ref
    d = dev()
    cfg, targets, layers, _ = load_case("s200", torch.float32, "cpu")
    cfg = dict(cfg, iter_matcher=False)
    crit = build(cfg)
    t_d = [{k: v.to(d) for k, v in t.items()} for t in targets]
    l_d = [{k: [None if v is None else v.to(d).requires_grad_(True) for v in lst] for k, lst in layer.items()} for layer in layers]
    out = crit(as_pred(l_d), t_d)
    (out["seg_loss"] + out["inst_loss"]).backward()
    t64 = [{k: (v.double() if v.is_floating_point() else v) for k, v in t.items()} for t in targets]
    l64 = [{k: [None if v is None else v.double().requires_grad_(True) for v in lst] for k, lst in layer.items()} for layer in layers]
    ref = loss_ref.unified_criterion(as_pred(l64), t64, cfg)
    (ref["seg_loss"] + ref["inst_loss"]).backward()
    assert abs(float(out["inst_loss"].detach()) - float(ref["inst_loss"].detach())) < 2e-5 * abs(float(ref["inst_loss"].detach()))
    for a, b in zip(l_d, l64):
        for k in ("cls_preds", "masks", "centers", "sizes"):
            for x, y in zip(a[k], b[k]):
                if x is None:
                    continue
                gr = y.grad if y.grad is not None else torch.zeros_like(y)
                assert (x.grad.cpu().double() - gr).abs().max().item() <= 2e-5 * max(gr.abs().max().item(), 1e-6), k
    # every layer used the last layer's match matrices
    for layer_matches in crit.last_matches[1:]:
        for m, m_last in zip(layer_matches, crit.last_matches[0]):
            assert m is m_last
