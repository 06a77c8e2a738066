"""Evaluation entry point (the reference's eval.py / run_voc_mAP.py surface, both stale upstream:
eval.py:14 imports a missing ``LossModel``; run_voc_mAP.py:74 passes a DataLoader to a per-sample loop).

Loads a checkpoint the way eval.py:63-68 does -- ``nn.DataParallel`` key prefix ``module.`` accepted -- and
runs the batched ``run_test_mAP`` (utils/utils.py:389-418).  Without a VOC list file the synthetic dataset is
used (smoke / throughput only: random weights give mAP ~ 0).
"""
import argparse
import time


from .utils.utils import prep_test_data, run_test_mAP
from .utils.YOLODataLoader import yoloDataset


def load_checkpoint(net, path, device):
    from . import checkpoint
    return checkpoint.load(net, path, device)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="resnet", choices=["resnet", "densenet"])
    ap.add_argument("--S", type=int, default=7)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--list-file", default=None, help="VOC image list (labels next to the images); default: synthetic")
    ap.add_argument("--num", type=int, default=256)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--fp8", action="store_true", help="ResNet only: run the forward on the fp8 (e4m3) MFMA executor")
    args = ap.parse_args(argv)
    device = "cuda:0"
    if args.backbone == "resnet":
        from .backbones.OriginResNet import resnet50
        net = resnet50(S=args.S)
    else:
        from .backbones.OriginDenseNet import densenet121
        net = densenet121(S=args.S)
    net = net.to(device)
    if args.checkpoint:
        load_checkpoint(net, args.checkpoint, device)
    net.eval()
    model = net
    if args.fp8:
        if args.backbone != "resnet":
            raise SystemExit("--fp8: the fp8 executor covers the ResNet backbone")
        from .infer_fp8 import ResNetFp8
        model = ResNetFp8(net)
    if args.list_file:
        ds = yoloDataset(args.list_file, train=False, with_file_path=True, S=args.S)
        target = prep_test_data(args.list_file, little_test=args.num)
    else:
        ds = yoloDataset(None, train=False, with_file_path=True, S=args.S, length=args.num)
        target = ds.synthetic_ground_truth()
    t0 = time.perf_counter()
    m = run_test_mAP(model, target, ds, len(ds), S=args.S, device=device, little_test=args.num, batch_size=args.batch_size)
    dt = time.perf_counter() - t0
    print("mAP %.5f over %d images, %.1f img/s (batched forward + GPU decoder/NMS)" % (m, min(args.num, len(ds)), min(args.num, len(ds)) / dt))


if __name__ == "__main__":
    main()
