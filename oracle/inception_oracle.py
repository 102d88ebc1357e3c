"""CPU fp32 restatement of the reference's InceptionV3 forward (test infrastructure).

TOPOLOGY PINNED, WEIGHTS UNPINNED.  Round 3: the graph this module executes is compared node for node (94 convs:
kernel / stride / padding / channels / sizes, BatchNorm eps, pools, concat order, 5 711 168 096 MAC) with the listing
of the same network that the reference repository ships, executed by path under a stub tensorflow
(``image_realism/IS/bird/inception/slim/inception_model.py:48-299`` -> ``tests/golden/inception_v3_topology.json``,
``tests/test_topology.py``).  What stays unpinned is the NUMBERS of a pretrained forward: the arithmetic behind
``image_realism/FID/inception.py:57`` (``torchvision.models.inception_v3(pretrained=True)``, torchvision==0.9.1 per
``requirements.txt:126``) is third-party and absent from /root/reference, and the build container has neither
torchvision nor the pretrained file.  This module restates torchvision's published Inception3 graph functionally from a
torchvision-format ``state_dict`` -- unfolded conv -> batch_norm(eps=1e-3) -> relu, NCHW, fp32, on the CPU.  Beside the
topology listing above, the reference's call sites (``inception.py:59-95`` block cuts, ``:117-124`` input handling) and
the parameter count of the published model (27 161 264) are checked by the same tests.

It deliberately shares no code with ``tise_toolbox_amd/inception.py`` (module
tree, BN folded, channels-last, GPU) so the two can be compared.
"""
import torch
import torch.nn.functional as F


class MacCounter:
    def __init__(self):
        self.macs = 0
        self.convs = 0


def _cbr(sd, name, x, stride=1, padding=0, counter=None):
    w = sd[name + ".conv.weight"]
    y = F.conv2d(x, w, None, stride, padding)
    if counter is not None:
        counter.convs += 1
        counter.macs += y.shape[1] * y.shape[2] * y.shape[3] * w.shape[1] * w.shape[2] * w.shape[3]
    y = F.batch_norm(y, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"],
                     sd[name + ".bn.weight"], sd[name + ".bn.bias"], False, 0.0, 0.001)
    return F.relu(y)


def _inception_a(sd, p, x, c):
    b1 = _cbr(sd, p + ".branch1x1", x, counter=c)
    b5 = _cbr(sd, p + ".branch5x5_2", _cbr(sd, p + ".branch5x5_1", x, counter=c), padding=2, counter=c)
    b3 = _cbr(sd, p + ".branch3x3dbl_1", x, counter=c)
    b3 = _cbr(sd, p + ".branch3x3dbl_2", b3, padding=1, counter=c)
    b3 = _cbr(sd, p + ".branch3x3dbl_3", b3, padding=1, counter=c)
    bp = _cbr(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1), counter=c)
    return torch.cat([b1, b5, b3, bp], 1)


def _inception_b(sd, p, x, c):
    b3 = _cbr(sd, p + ".branch3x3", x, stride=2, counter=c)
    bd = _cbr(sd, p + ".branch3x3dbl_1", x, counter=c)
    bd = _cbr(sd, p + ".branch3x3dbl_2", bd, padding=1, counter=c)
    bd = _cbr(sd, p + ".branch3x3dbl_3", bd, stride=2, counter=c)
    return torch.cat([b3, bd, F.max_pool2d(x, 3, 2)], 1)


def _inception_c(sd, p, x, c):
    b1 = _cbr(sd, p + ".branch1x1", x, counter=c)
    b7 = _cbr(sd, p + ".branch7x7_1", x, counter=c)
    b7 = _cbr(sd, p + ".branch7x7_2", b7, padding=(0, 3), counter=c)
    b7 = _cbr(sd, p + ".branch7x7_3", b7, padding=(3, 0), counter=c)
    bd = _cbr(sd, p + ".branch7x7dbl_1", x, counter=c)
    bd = _cbr(sd, p + ".branch7x7dbl_2", bd, padding=(3, 0), counter=c)
    bd = _cbr(sd, p + ".branch7x7dbl_3", bd, padding=(0, 3), counter=c)
    bd = _cbr(sd, p + ".branch7x7dbl_4", bd, padding=(3, 0), counter=c)
    bd = _cbr(sd, p + ".branch7x7dbl_5", bd, padding=(0, 3), counter=c)
    bp = _cbr(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1), counter=c)
    return torch.cat([b1, b7, bd, bp], 1)


def _inception_d(sd, p, x, c):
    b3 = _cbr(sd, p + ".branch3x3_2", _cbr(sd, p + ".branch3x3_1", x, counter=c), stride=2, counter=c)
    b7 = _cbr(sd, p + ".branch7x7x3_1", x, counter=c)
    b7 = _cbr(sd, p + ".branch7x7x3_2", b7, padding=(0, 3), counter=c)
    b7 = _cbr(sd, p + ".branch7x7x3_3", b7, padding=(3, 0), counter=c)
    b7 = _cbr(sd, p + ".branch7x7x3_4", b7, stride=2, counter=c)
    return torch.cat([b3, b7, F.max_pool2d(x, 3, 2)], 1)


def _inception_e(sd, p, x, c):
    b1 = _cbr(sd, p + ".branch1x1", x, counter=c)
    b3 = _cbr(sd, p + ".branch3x3_1", x, counter=c)
    b3 = torch.cat([_cbr(sd, p + ".branch3x3_2a", b3, padding=(0, 1), counter=c),
                    _cbr(sd, p + ".branch3x3_2b", b3, padding=(1, 0), counter=c)], 1)
    bd = _cbr(sd, p + ".branch3x3dbl_1", x, counter=c)
    bd = _cbr(sd, p + ".branch3x3dbl_2", bd, padding=1, counter=c)
    bd = torch.cat([_cbr(sd, p + ".branch3x3dbl_3a", bd, padding=(0, 1), counter=c),
                    _cbr(sd, p + ".branch3x3dbl_3b", bd, padding=(1, 0), counter=c)], 1)
    bp = _cbr(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1), counter=c)
    return torch.cat([b1, b3, bd, bp], 1)


@torch.no_grad()
def inception_forward(sd, x, last_block=3, resize_input=True, normalize_input=True, counter=None):
    """image_realism/FID/inception.py:100-134 -> list of the 4 block outputs (up to last_block).

    ``sd``: torchvision-format state_dict (fp32 CPU tensors); ``x``: (B,3,H,W) fp32 in [0,1].
    """
    c = counter
    x = x.float()
    if resize_input:                                              # :117-118
        x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=True)
    if normalize_input:                                           # :120-124
        x = x.clone()
        x[:, 0] = x[:, 0] * (0.229 / 0.5) + (0.485 - 0.5) / 0.5
        x[:, 1] = x[:, 1] * (0.224 / 0.5) + (0.456 - 0.5) / 0.5
        x[:, 2] = x[:, 2] * (0.225 / 0.5) + (0.406 - 0.5) / 0.5
    outs = []
    # block 0 (:59-66)
    x = _cbr(sd, "Conv2d_1a_3x3", x, stride=2, counter=c)
    x = _cbr(sd, "Conv2d_2a_3x3", x, counter=c)
    x = _cbr(sd, "Conv2d_2b_3x3", x, padding=1, counter=c)
    x = F.max_pool2d(x, 3, 2)
    outs.append(x)
    if last_block >= 1:                                           # :68-71
        x = _cbr(sd, "Conv2d_3b_1x1", x, counter=c)
        x = _cbr(sd, "Conv2d_4a_3x3", x, counter=c)
        x = F.max_pool2d(x, 3, 2)
        outs.append(x)
    if last_block >= 2:                                           # :73-85
        x = _inception_a(sd, "Mixed_5b", x, c)
        x = _inception_a(sd, "Mixed_5c", x, c)
        x = _inception_a(sd, "Mixed_5d", x, c)
        x = _inception_b(sd, "Mixed_6a", x, c)
        x = _inception_c(sd, "Mixed_6b", x, c)
        x = _inception_c(sd, "Mixed_6c", x, c)
        x = _inception_c(sd, "Mixed_6d", x, c)
        x = _inception_c(sd, "Mixed_6e", x, c)
        outs.append(x)
    if last_block >= 3:                                           # :87-95
        x = _inception_d(sd, "Mixed_7a", x, c)
        x = _inception_e(sd, "Mixed_7b", x, c)
        x = _inception_e(sd, "Mixed_7c", x, c)
        x = F.adaptive_avg_pool2d(x, (1, 1))
        outs.append(x)
    return outs


@torch.no_grad()
def logits_from_pool3(sd, pool3, bias=True):
    """torchvision Inception3 classifier head: fc(flatten(pool3)) (dropout is identity in eval).

    ``bias=False``: the head of image_realism/IS/coco/inception_score_star_coco.py:104-105, which multiplies pool3 by
    the weight matrix of the last layer only (``w = ...("softmax/logits/MatMul").inputs[1]; logits = tf.matmul(
    tf.squeeze(pool3, [1, 2]), w)``; the graph's BiasAdd is not applied) before tf.div(logits, T) (:107).
    ``bias=True``: object_centric_inception_score.py:41-60 (the whole torch model) and the slim ``logits`` end point
    of inception_score_star_bird.py:189."""
    return F.linear(pool3.flatten(1).float(), sd["fc.weight"], sd["fc.bias"] if bias else None)
