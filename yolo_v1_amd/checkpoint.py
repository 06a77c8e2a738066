"""Checkpoint I/O compatible with the reference's files.

* ``init_from_pretrained`` -- the by-name initialisation of train.py:59-78: every entry of an ImageNet
  (torchvision-keyed) state_dict whose key exists in the YOLO backbone and does not start with ``fc`` is taken;
  everything else (the YOLO stages, ``layer6``, ``bn_end``) keeps its fresh initialisation.
* ``save`` -- writes ``state_dict()`` with the ``module.`` prefix ``nn.DataParallel`` gives the reference's files
  (train.py:80,:207-209), fp32 OIHW tensors, so reference eval.py:66 loads them unchanged.
* ``load`` -- accepts files with or without the prefix (eval.py:63-68); ``weights_only=True`` (nothing from the
  file is executed).
On disk everything is fp32 OIHW as torch keeps it; the KRSC bf16 shadows the kernels read are rebuilt from the
parameters after a load (the parameter version counters change, see ops.ConvWeights).
"""
import torch


def strip_module_prefix(sd):
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}


def init_from_pretrained(net, pretrained_sd):
    """Returns the list of keys that were taken from ``pretrained_sd``."""
    dd = net.state_dict()
    taken = []
    for k, v in strip_module_prefix(pretrained_sd).items():
        if k in dd and not k.startswith('fc'):
            if tuple(dd[k].shape) != tuple(v.shape):
                raise ValueError("pretrained tensor %s has shape %s, the backbone expects %s"
                                 % (k, tuple(v.shape), tuple(dd[k].shape)))
            dd[k] = v
            taken.append(k)
    net.load_state_dict(dd)
    _weights_changed()
    return taken


def save(net, path, data_parallel_prefix=True):
    sd = net.state_dict()
    if data_parallel_prefix:
        sd = {'module.' + k: v for k, v in sd.items()}
    torch.save({k: v.detach().to('cpu').contiguous() for k, v in sd.items()}, path)


def load(net, path, device=None, strict=True):
    sd = torch.load(path, map_location=device or 'cpu', weights_only=True)
    out = net.load_state_dict(strip_module_prefix(sd), strict=strict)
    _weights_changed()
    return out


def _weights_changed():
    from . import ops
    ops.bump_weight_epoch()
