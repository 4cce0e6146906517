"""One-time model preparation the reference does before quantizing:
BN folding (utils/absorb_bn.py:5-41), marking the convolutions that feed a ReLU so they are
quantized half-range (utils/mark_relu.py:4-29) and tensorboard-style node names
(utils/model_naming.py:4-28).  Architecture-agnostic: the marks are derived from the block
structure of the in-repo models (harness/models.py) instead of torchvision classes."""
import torch
import torch.nn as nn


def fold_bn_into(conv, bn):
    """conv <- conv followed by bn (inference statistics); bn becomes the identity."""
    w = conv.weight.data
    if conv.bias is None:
        conv.bias = nn.Parameter(torch.zeros(w.shape[0], dtype=w.dtype, device=w.device))
    b = conv.bias.data
    invstd = (bn.running_var + bn.eps).rsqrt()
    w.mul_(invstd.view(-1, *([1] * (w.dim() - 1))))
    b.sub_(bn.running_mean).mul_(invstd)
    if bn.affine:
        w.mul_(bn.weight.data.view(-1, *([1] * (w.dim() - 1))))
        b.mul_(bn.weight.data).add_(bn.bias.data)
    bn.running_mean.zero_()
    bn.running_var.fill_(1.)
    bn.register_parameter('weight', None)
    bn.register_parameter('bias', None)
    bn.affine = False
    bn.eps = 0.
    bn.absorbed = True


def absorb_bn(model):
    """Fold every BatchNorm that directly follows a Conv2d (groups == 1) or Linear among the
    children of the same module, recursively."""
    prev = None
    for m in model.children():
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)) and prev is not None and \
                ((isinstance(prev, nn.Conv2d) and prev.groups == 1) or isinstance(prev, nn.Linear)):
            fold_bn_into(prev, m)
        absorb_bn(m)
        prev = m


def set_node_names(model, root=None):
    """`internal_name` on every leaf, e.g. ResNet/Sequential[layer1]/Bottleneck[0]/Conv2d[conv1]."""
    def tname(m):
        return type(m).__name__.replace('WithId', '')

    def walk(parent, name):
        kids = list(parent.named_children())
        if not kids:
            parent.internal_name = name
        for n, m in kids:
            walk(m, '%s/%s[%s]' % (name, tname(m), n))
    walk(model, root or tname(model))
