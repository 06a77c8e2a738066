"""CPU: pin the oracle (oracle/*) against golden vectors produced by the
reference's own modules (oracle/gen_golden.py) and against the two known-answer
snippets the reference ships."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_cases
from oracle import backbones as ob
from oracle import boxes as obx
from oracle import loss as ol
from oracle import voc as ov


# ---------------------------------------------------------------- IoU / convert
def test_iou_known_answer_reference_main():
    # the reference's own __main__ smoke, utils/utils.py:506-525
    z = np.load(os.path.join(GOLDEN, "iou_cases.npz"))
    got = obx.compute_iou_matrix(z["known_b1"], z["known_b2"])
    np.testing.assert_array_equal(got, z["known_iou"])
    np.testing.assert_allclose(got, [[0.2445, 0.5383, 0.0], [0.0, 0.0, 0.1701]], atol=5e-5)


def test_iou_random_and_convert_bit_exact():
    z = np.load(os.path.join(GOLDEN, "iou_cases.npz"))
    np.testing.assert_array_equal(obx.compute_iou_matrix(z["r1"], z["r2"]), z["r_iou"])
    np.testing.assert_array_equal(obx.convert_cxcywh_to_x1y1x2y2(z["cx"], 7), z["cx7"])
    np.testing.assert_array_equal(obx.convert_cxcywh_to_x1y1x2y2(z["cx"], 14), z["cx14"])


# ---------------------------------------------------------------- encoder
def test_encoder_bit_exact():
    for c in load_cases("encoder_cases.npz"):
        got = obx.encode_target(c["boxes"], c["labels"], int(c["S"]))
        np.testing.assert_array_equal(got, c["target"])


# ---------------------------------------------------------------- loss
@pytest.mark.parametrize("case", load_cases("loss_cases.npz"), ids=lambda c: "S%d_N%d_%s" % (c["S"], c["N"], c["kind"]))
def test_loss_value_components_grad(case):
    S, bs = int(case["S"]), int(case["bs"])
    loss, comps, grad = ol.yolo_loss_and_grad(case["pred"], case["target"], S, 2, 20, 5.0, 0.5, bs)
    # tolerance: fp32 kernels vs reference <= 1e-5 rel / 1e-6 abs
    np.testing.assert_allclose(loss, case["loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(comps, case["comps"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(grad, case["grad"], rtol=1e-5, atol=1e-6)


def test_loss_accepts_permuted_pred():
    c = load_cases("loss_cases.npz")[4]
    S = int(c["S"])
    p = torch.tensor(c["pred"]).permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)   # as OriginResNet.py:189
    assert not p.is_contiguous()
    total, _ = ol.yolo_loss(p, torch.tensor(c["target"]), S, 2, 20, batch_size=int(c["bs"]))
    np.testing.assert_allclose(float(total), c["loss"], rtol=1e-5)


# ---------------------------------------------------------------- NMS / decoder
def test_nms_bit_exact_vs_unmodified_reference():
    cases = load_cases("nms_cases.npz")
    assert len(cases) >= 50
    for c in cases:
        keep = obx.nms(c["boxes"], c["scores"], float(c["thr"]))
        np.testing.assert_array_equal(keep, c["keep"])


def test_nms_single_survivor_keeps_last_and_stops():
    # the torch>=0.5 crash case: 0.4 semantics = keep the lone survivor
    b = np.array([[0, 0, 1, 1], [0, 0, 1, 1.01], [2, 2, 3, 3]], np.float32)
    s = np.array([0.9, 0.8, 0.7], np.float32)
    np.testing.assert_array_equal(obx.nms(b, s, 0.5), [0, 2])
    np.testing.assert_array_equal(obx.nms(b[:1], s[:1], 0.5), [0])
    np.testing.assert_array_equal(obx.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5), [])


def test_decoder_candidates_and_full():
    for c in load_cases("decoder_cases.npz"):
        S = int(c["S"])
        bx, cl, pr, _ = obx.decode_candidates(c["pred"][0], S, 2, float(c["thresh"]))
        if bx.shape[0] == 0:       # reference substitutes one zero box (utils.py:134-137)
            assert c["cand_boxes"].shape == (1, 4) and not c["cand_boxes"].any()
        else:
            np.testing.assert_array_equal(bx, c["cand_boxes"])
            np.testing.assert_array_equal(cl, c["cand_cls"])
            np.testing.assert_array_equal(pr, c["cand_probs"])
        if int(c["full_ok"]):
            fb, fc, fp, _ = obx.decoder(c["pred"], S, 2, float(c["thresh"]), float(c["nms_th"]))
            np.testing.assert_array_equal(fb, c["full_boxes"])
            np.testing.assert_array_equal(fc, c["full_cls"])
            np.testing.assert_array_equal(fp, c["full_probs"])


# ---------------------------------------------------------------- VOC AP
def test_voc_eval_known_answer():
    k = json.load(open(os.path.join(GOLDEN, "voc_eval_known.json")))
    target = {tuple(s.split("|")): v for s, v in k["target"].items()}
    m, aps = ov.voc_eval(k["preds"], target, k["classes"])
    assert abs(m - 0.9166666666666666) < 1e-12 and abs(m - k["mAP"]) < 1e-12     # utils/utils.py:321-324
    assert abs(aps[0] - 0.8333333333) < 1e-9 and aps[1] == 1.0
    m2, _ = ov.voc_eval(k["preds"], target, k["classes_quirk"])                  # -1 then break, :248-255
    assert abs(m2 - k["mAP_quirk"]) < 1e-12
    assert abs(ov.voc_ap(np.array(k["rec"]), np.array(k["prec"])) - k["ap_area"]) < 1e-12
    assert abs(ov.voc_ap(np.array(k["rec"]), np.array(k["prec"]), True) - k["ap_07"]) < 1e-12


# ---------------------------------------------------------------- backbone blocks
def _block(name):
    z = np.load(os.path.join(GOLDEN, "block_cases.npz"))
    pre = name + "/"
    d = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    P = {k[2:]: torch.tensor(v) for k, v in d.items() if k.startswith("p/")}
    return d, P


@pytest.mark.parametrize("name,fn", [
    ("bneck_s1_ds", lambda x, P: ob.bottleneck(x, P, "", 1)),
    ("bneck_s2_ds", lambda x, P: ob.bottleneck(x, P, "", 2)),
    ("bneck_plain", lambda x, P: ob.bottleneck(x, P, "", 1)),
    ("bneck_s2_ds_16", lambda x, P: ob.bottleneck(x, P, "", 2)),
    ("bneck_plain_32", lambda x, P: ob.bottleneck(x, P, "", 1)),
    ("dense_layer", lambda x, P: ob.dense_layer(x, P, "")),
    ("transition", lambda x, P: ob.transition(x, P, "")),
])
def test_block_forward_backward_vs_reference_modules(name, fn):
    d, P = _block(name)
    P = {"." + k: v for k, v in P.items()}          # oracle prefixes with p + ".conv1" etc; p == ""
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    x = torch.tensor(d["x"], requires_grad=True)
    y = fn(x, P)
    np.testing.assert_allclose(y.detach().numpy(), d["y"], rtol=1e-5, atol=1e-5)
    y.backward(torch.tensor(d["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), d["gx"], rtol=1e-4, atol=1e-5)
    for k in d:
        if k.startswith("g/"):
            np.testing.assert_allclose(P["." + k[2:]].grad.numpy(), d[k], rtol=1e-4, atol=2e-5, err_msg=k)
        if k.startswith("after/"):
            np.testing.assert_allclose(P["." + k[6:]].detach().numpy(), d[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_state_dict_inventory_matches_reference():
    inv = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["inventory"]
    for kind, fn in (("resnet", ob.resnet50_param_shapes), ("densenet", ob.densenet121_param_shapes)):
        for S in (7, 14):
            mine = [[k, list(v)] for k, v in fn(S).items()]
            assert mine == inv["%s_S%d" % (kind, S)]
    n = sum(int(np.prod(s)) for k, s in ob.resnet50_param_shapes(7).items() if "running" not in k and "num_batches" not in k)
    assert n == 41155708


@pytest.mark.parametrize("kind,S", [("resnet", 7), ("resnet", 14), ("densenet", 7), ("densenet", 14)])
def test_wholenet_forward_vs_reference(kind, S):
    z = np.load(os.path.join(GOLDEN, "wholenet_fwd.npz"))
    shapes = ob.resnet50_param_shapes(S) if kind == "resnet" else ob.densenet121_param_shapes(S)
    P = ob.init_params(shapes, kind, seed=S)
    x = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(100 + S))
    fwd = ob.resnet50_forward if kind == "resnet" else ob.densenet121_forward
    with torch.no_grad():
        y = fwd(x, P, S, training=True)
    np.testing.assert_allclose(y.numpy(), z["%s_S%d_y" % (kind, S)], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(P["bn_end.running_mean"].numpy(), z["%s_S%d_bn_end_rm" % (kind, S)], rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------- train-step glue (a11)
def test_lr_policy():
    from oracle import train_step as ots
    lr = 0.0
    for it in range(1, 1003):
        lr = ots.learning_rate_policy(it, 0, lr, {1: 0.001, 75: 0.0001})
    assert abs(lr - 1000 * 1e-6) < 1e-12          # warm-up stops after 1000 iterations (train.py:22-25)
    assert ots.learning_rate_policy(5, 1, 0.5, {1: 0.001}) == 0.001


@pytest.mark.parametrize("tag,epoch,lr_map", [("warmup", 0, {}), ("epoch1", 1, {1: 0.001})])
def test_train_steps_vs_reference_modules(tag, epoch, lr_map):
    from oracle import train_step as ots
    ref = json.load(open(os.path.join(GOLDEN, "train_steps.json")))[tag]
    P = ots.make_state("resnet", 7, seed=0)
    images, target = ots.synthetic_batch(2, 7)
    got = ots.train_steps(P, images, target, 7, 3, "resnet", epoch=epoch, lr_map=lr_map)
    for g, r in zip(got, ref):
        assert abs(g["lr"] - r["lr"]) < 1e-15
        # N=2 batch-norm statistics make the trajectory sensitive; fp32 re-association between
        # module and functional calls stays well inside 1e-3
        np.testing.assert_allclose(g["loss"], r["loss"], rtol=1e-3)
        np.testing.assert_allclose(g["comps"], r["comps"], rtol=2e-3, atol=1e-4)
