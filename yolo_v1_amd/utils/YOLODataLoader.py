"""Drop-in for the reference's ``utils/YOLODataLoader.py`` surface that the hot path needs.

``yoloDataset`` keeps the constructor signature and the ``__getitem__`` tuple
(``img[3,448,448], target[S,S,B*5+C][, fname]``, reference :13, :156-194) and the target
``encoder`` (:200-230).  Two sources:
  * ``list_file=None`` (or ``synthetic=True``): VOC-shaped synthetic samples -- images
    ``randn(3,448,448)``, ``objs`` boxes with cx,cy~U(0,1), w,h~U(0.05,0.9), class~U{0..C-1}
    (SURVEY 8d); nothing is read from disk, so the throughput bench needs no dataset;
  * a darknet-style label list (``<img path>`` lines, labels next to the images as
    ``cls cx cy w h`` rows, reference :94-106): boxes/labels are read and encoded; images are decoded,
    resized and normalised by ``pil_image_loader`` (or any ``image_loader`` callable); the imgaug
    augmentation pipeline (:31-79, :161-172) needs cv2/imgaug and is out of scope here.
``encode_boxes`` is host code, as in the reference (it runs in DataLoader worker processes);
``encode_targets_device`` is the same encoder as one HIP kernel over a whole batch, and
``DevicePrefetcher`` feeds the training loop: pinned staging buffers, H2D copies and the device encoder on a
copy stream, one batch ahead of the step that is running.
"""

import torch
import torch.utils.data as data


def encode_boxes(boxes, labels, S, B=2, C=20):
    """[k,4] normalised (cx,cy,w,h) + [k] labels -> target [S,S,B*5+C] (reference :200-230).

    Cell = ceil(c / (1/S)) - 1 per axis (row = y, col = x); a later box landing in an occupied cell
    replaces it; both confidence slots are 1; the box (offset inside the cell, w, h) fills every
    slot; one-hot class.
    """
    D = B * 5 + C
    target = torch.zeros((S, S, D))
    boxes = torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 4)
    if boxes.shape[0] == 0:
        return target
    cell = torch.tensor(1.0 / S, dtype=torch.float32)
    ij = torch.ceil(boxes[:, :2] / cell) - 1            # [k,2] (col, row)
    delta = (boxes[:, :2] - ij * cell) / cell
    for k in range(boxes.shape[0]):
        col, row = int(ij[k, 0]), int(ij[k, 1])
        cellv = torch.zeros(D)
        cellv[:B] = 1
        cellv[B * 5 + int(labels[k])] = 1
        cellv[B:B * 5] = torch.cat([delta[k], boxes[k, 2:]]).repeat(B)
        target[row, col] = cellv
    return target


IMAGENET_MEAN = (0.485, 0.456, 0.406)      # train.py:108 (applied to the BGR channels cv2.imread returns, as the reference does)
IMAGENET_STD = (0.229, 0.224, 0.225)


def pil_image_loader(path, size=448, bgr=True):
    """Image file -> normalised fp32 tensor [3,size,size]: what the reference's transform (train.py:105-109:
    ``cv_resize`` -> ``ToTensor`` -> ``Normalize``) makes of ``cv2.imread(path)``, with PIL standing in for cv2 (bilinear
    resize; not bit-identical to cv2's -- image decoding is outside the pinned path).  No augmentation."""
    from PIL import Image
    import numpy as np
    with Image.open(path) as im:
        im = im.convert("RGB").resize((size, size), Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32) / 255.0
    if bgr:
        a = a[:, :, ::-1]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (t - mean) / std


def collate_raw(samples):
    """DataLoader ``collate_fn`` for ``yoloDataset(raw_targets=True)``: images stacked, boxes/labels padded to the
    largest count of the batch -> (images [N,3,H,W], boxes [N,Kmax,4], labels [N,Kmax] int64, counts [N] int32[, fnames])."""
    imgs = torch.stack([s[0] for s in samples])
    counts = torch.tensor([int(s[1].shape[0]) for s in samples], dtype=torch.int32)
    kmax = max(1, int(counts.max()) if len(samples) else 1)
    boxes = torch.zeros((len(samples), kmax, 4), dtype=torch.float32)
    labels = torch.zeros((len(samples), kmax), dtype=torch.int64)
    for i, s in enumerate(samples):
        k = int(counts[i])
        if k:
            boxes[i, :k] = torch.as_tensor(s[1], dtype=torch.float32).reshape(-1, 4)
            labels[i, :k] = torch.as_tensor(s[2], dtype=torch.int64).reshape(-1)
    out = (imgs, boxes, labels, counts)
    if len(samples) and len(samples[0]) > 3:
        out = out + ([s[3] for s in samples],)
    return out


def encode_targets_device(boxes, labels, counts, S, B=2, C=20, out=None, check=True):
    """The reference encoder (:200-230) for a whole batch on the GPU: boxes [N,K,4] fp32, labels [N,K] int64,
    counts [N] int32 (device tensors) -> target [N,S,S,B*5+C] fp32, bit-identical to ``encode_boxes`` per image.
    ``check`` reads back the out-of-grid flag (one host sync) and raises IndexError like the reference would."""
    from .._lib import check as _check, lib, ptr, require_cuda, stream_ptr
    require_cuda(boxes, labels, counts)
    N, K = int(boxes.shape[0]), int(boxes.shape[1])
    if boxes.dtype != torch.float32 or labels.dtype != torch.int64 or counts.dtype != torch.int32:
        raise TypeError("encode_targets_device: boxes fp32, labels int64, counts int32")
    if tuple(labels.shape) != (N, K) or tuple(boxes.shape) != (N, K, 4) or tuple(counts.shape) != (N,):
        raise ValueError("encode_targets_device: boxes [N,K,4], labels [N,K], counts [N]")
    boxes, labels, counts = boxes.contiguous(), labels.contiguous(), counts.contiguous()
    if out is None:
        out = torch.empty((N, S, S, B * 5 + C), dtype=torch.float32, device=boxes.device)
    elif tuple(out.shape) != (N, S, S, B * 5 + C) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("encode_targets_device: out must be a contiguous fp32 [N,S,S,B*5+C] tensor")
    err = torch.empty(1, dtype=torch.int32, device=boxes.device)
    _check(lib().yv1_encode_targets(ptr(boxes), ptr(labels), ptr(counts), N, K, S, B, C, ptr(out), ptr(err),
                                    stream_ptr(boxes.device)), "yv1_encode_targets")
    if check and int(err.item()) != 0:
        raise IndexError("encode_targets_device: a box centre falls outside the %dx%d grid or a label outside [0,%d)"
                         % (S, S, C))
    return out


class DevicePrefetcher:
    """Wraps an iterable of host batches -- ``(images, boxes, labels, counts)`` from ``collate_raw`` or
    ``(images, target)`` -- and yields device ``(images, target)`` pairs.  Batch t+1 is staged through pinned
    memory, copied and encoded on a copy stream while step t runs; the consumer's stream waits on an event, never
    on the host.  Two staging slots, so a pinned buffer is not rewritten while its copy may still be in flight."""

    def __init__(self, loader, device, S, B=2, C=20, out_images=None, out_target=None):
        self.loader, self.device, self.S, self.B, self.C = loader, torch.device(device), S, B, C
        self.copy_stream = torch.cuda.Stream(self.device)
        self.static = (out_images, out_target)      # optional static destination buffers (hipGraph replay)
        self._pinned = [{}, {}]
        self._slot_done = [None, None]

    def _pin(self, slot, name, t):
        buf = self._pinned[slot].get(name)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._pinned[slot][name] = buf
        buf.copy_(t)
        return buf

    def _stage(self, batch, slot):
        if self._slot_done[slot] is not None:
            self._slot_done[slot].synchronize()      # the copy that last read this slot's pinned buffers is done
        names = ("images", "boxes", "labels", "counts") if len(batch) >= 4 else ("images", "target")
        host = [self._pin(slot, n, t) for n, t in zip(names, batch)]
        with torch.cuda.stream(self.copy_stream):
            dev = [h.to(self.device, non_blocking=True) for h in host]
            done = torch.cuda.Event()
            done.record(self.copy_stream)
            self._slot_done[slot] = done
            if len(dev) >= 4:
                target = encode_targets_device(dev[1], dev[2], dev[3], self.S, self.B, self.C, check=False)
            else:
                target = dev[1]
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return dev[0], target, ready

    def __iter__(self):
        it = iter(self.loader)
        slot = 0
        try:
            nxt = self._stage(next(it), slot)
        except StopIteration:
            return
        while nxt is not None:
            images, target, ready = nxt
            slot ^= 1
            try:
                nxt = self._stage(next(it), slot)
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            images.record_stream(cur)
            target.record_stream(cur)
            if self.static[0] is not None:
                self.static[0].copy_(images)
                self.static[1].copy_(target)
                yield self.static
            else:
                yield images, target


class yoloDataset(data.Dataset):
    image_size = 448

    def __init__(self, list_file=None, train=True, transform=None, device='cpu', little_train=False,
                 with_file_path=False, S=7, B=2, C=20, test_mode=False, synthetic=None, length=512, objs=3,
                 seed=1234, image_loader=None, image_size=None, raw_targets=False):
        self.raw_targets = raw_targets            # True: return (img, boxes, labels): the batch is encoded on the GPU
        self.train = train
        self.transform = transform
        self.S, self.B, self.C = S, B, C
        self.device = device
        self.with_file_path = with_file_path
        self._test = test_mode
        self.synthetic = (list_file is None) if synthetic is None else synthetic
        self.objs = objs
        self.seed = seed
        self.image_loader = image_loader
        if image_size is not None:
            self.image_size = image_size
        self.fnames = []
        if not self.synthetic:
            with open(list_file) as f:
                lines = f.readlines()
            if little_train:
                lines = lines[:64 * 8]
            self.fnames = [ln.strip().split()[0] for ln in lines if ln.strip()]
            self.num_samples = len(self.fnames)
        else:
            self.num_samples = length

    def __len__(self):
        return self.num_samples

    def encoder(self, boxes, labels):
        return encode_boxes(boxes, labels, self.S, self.B, self.C)

    @staticmethod
    def get_boxes_labels(in_path):
        boxes, labels = [], []
        with open(in_path.replace('JPEGImages', 'labels').replace('jpg', 'txt'), 'r') as f:
            for line in f:
                ll = line.strip().split(' ')
                labels.append(int(ll[0]))
                boxes.append([float(v) for v in ll[1:5]])
        return torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4), torch.tensor(labels, dtype=torch.long)

    def _synthetic_item(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        img = torch.randn(3, self.image_size, self.image_size, generator=g)
        cxcy = torch.rand(self.objs, 2, generator=g).clamp_(1e-3, 1.0)
        wh = torch.rand(self.objs, 2, generator=g) * 0.85 + 0.05
        labels = torch.randint(0, self.C, (self.objs,), generator=g)
        return img, torch.cat([cxcy, wh], 1), labels, "synthetic_%06d.jpg" % idx

    def synthetic_ground_truth(self, n=None):
        """voc_eval target dict {(image_id, class_name): [[x0,y0,x1,y1], ...]} of the synthetic samples, built the
        way the reference builds it from label files (utils/utils.py:326-345,:356-387)."""
        from collections import defaultdict
        from .utils import VOC_CLASSES
        target = defaultdict(list)
        sz = self.image_size
        for idx in range(self.num_samples if n is None else n):
            _, boxes, labels, fname = self._synthetic_item(idx)
            for (x, y, w, h), lab in zip(boxes.tolist(), labels.tolist()):
                target[(fname.split('.')[0], VOC_CLASSES[lab])].append(
                    [int((x - 0.5 * w) * sz), int((y - 0.5 * h) * sz), int((x + 0.5 * w) * sz), int((y + 0.5 * h) * sz)])
        return target

    def __getitem__(self, idx):
        if self.synthetic:
            img, boxes, labels, fname = self._synthetic_item(idx)
        else:
            fname = self.fnames[idx]
            loader = self.image_loader
            if loader is None:          # default: PIL decode + resize + normalise (no cv2 / imgaug in this package)
                loader = lambda f: pil_image_loader(f, self.image_size)
            img = loader(fname)
            if self.transform is not None:
                img = self.transform(img)
            boxes, labels = self.get_boxes_labels(fname)
        if self.raw_targets:
            return (img, boxes, labels, fname) if self.with_file_path else (img, boxes, labels)
        target = self.encoder(boxes, labels)
        if self.with_file_path:
            return img, target, fname
        return img, target


def synthetic_batch(N, S, B=2, C=20, seed=1234, objs=3, hw=448, device='cpu'):
    """A whole synthetic batch (images [N,3,hw,hw] fp32, targets [N,S,S,B*5+C]) for benches/smoke."""
    ds = yoloDataset(None, S=S, B=B, C=C, objs=objs, seed=seed, length=N)
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(N, 3, hw, hw, generator=g)
    targets = torch.stack([ds.encoder(*ds._synthetic_item(i)[1:3]) for i in range(N)])
    return images.to(device), targets.to(device)
