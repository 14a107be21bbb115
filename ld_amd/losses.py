"""The loss modules of the LD path with mmdet's registry names, constructor
arguments and ``forward(pred, target, weight, avg_factor, reduction_override)``
contract (mmdet/models/losses/{kd_loss,gfocal_loss,iou_loss,utils}.py).

Called on their own they run the reference-layout row kernels of
libldhip.so (rows.hip); inside LDHead the whole block is fused instead
(lossblock.py).  Reduction follows ``weight_reduce_loss``
(losses/utils.py:28-55) exactly.
"""
import torch
import torch.nn as nn

from . import lib as L
from .registry import LOSSES


def reduce_loss(loss, reduction):
    """losses/utils.py:7-25."""
    if reduction == 'none':
        return loss
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise ValueError(reduction)


def _reduce_weighted(loss_rows, reduction, avg_factor):
    """``loss_rows`` already carries the element-wise weight."""
    if avg_factor is None:
        return reduce_loss(loss_rows, reduction)
    if reduction == 'mean':
        return loss_rows.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss_rows


class _RowLoss(torch.autograd.Function):
    """One launch computes the per-row loss (weight applied) and d/dpred; the
    backward just scales the stored gradient by the incoming row grads."""

    @staticmethod
    def forward(ctx, launcher, pred, weight, *consts):
        pred = L.require_device(pred.contiguous(), torch.float32, 'pred')
        if weight is not None:
            weight = L.require_device(weight.contiguous().float(),
                                      torch.float32, 'weight')
        rows = pred.shape[0]
        loss_rows = pred.new_empty(rows)
        grad = torch.empty_like(pred) if ctx.needs_input_grad[1] else None
        if rows:
            launcher(pred, weight, loss_rows, grad, *consts)
        ctx.grad = grad
        ctx.nconst = len(consts)
        return loss_rows

    @staticmethod
    def backward(ctx, g_rows):
        g = ctx.grad
        if g is not None:
            g = g * g_rows.reshape(-1, *([1] * (g.dim() - 1)))
        return (None, g, None) + (None, ) * ctx.nconst


def _st(t):
    return L.stream_ptr(t.device)


def _launch_kl(pred, weight, loss_rows, grad, soft, T):
    soft = L.require_device(soft.detach().contiguous(), torch.float32, 'soft')
    L.check(L.get_lib().ld_kd_kl_rows(L.ptr(pred), L.ptr(soft), L.ptr(weight),
                                      pred.shape[0], pred.shape[1], float(T),
                                      1.0, L.ptr(loss_rows), L.ptr(grad),
                                      _st(pred)), 'ld_kd_kl_rows')


def _launch_qfl(pred, weight, loss_rows, grad, label, score):
    label = L.require_device(label.contiguous(), torch.int64, 'label')
    score = L.require_device(score.contiguous().float(), torch.float32,
                             'score')
    L.check(L.get_lib().ld_qfl_rows(L.ptr(pred), L.ptr(label), L.ptr(score),
                                    L.ptr(weight), pred.shape[0],
                                    pred.shape[1], 1.0, L.ptr(loss_rows),
                                    L.ptr(grad), _st(pred)), 'ld_qfl_rows')


def _launch_dfl(pred, weight, loss_rows, grad, target):
    target = L.require_device(target.contiguous().float(), torch.float32,
                              'target')
    L.check(L.get_lib().ld_dfl_rows(L.ptr(pred), L.ptr(target), L.ptr(weight),
                                    pred.shape[0], pred.shape[1], 1.0,
                                    L.ptr(loss_rows), L.ptr(grad), _st(pred)),
            'ld_dfl_rows')


def _launch_giou(pred, weight, loss_rows, grad, target, eps):
    target = L.require_device(target.contiguous(), torch.float32, 'target')
    L.check(L.get_lib().ld_giou_rows(L.ptr(pred), L.ptr(target), L.ptr(weight),
                                     pred.shape[0], float(eps), 1.0,
                                     L.ptr(loss_rows), L.ptr(grad), _st(pred)),
            'ld_giou_rows')


def knowledge_distillation_kl_div_loss(pred, soft_label, T, weight=None,
                                       reduction='mean', avg_factor=None,
                                       detach_target=True):
    """kd_loss.py:10-36 (+ the weighted_loss wrapper)."""
    assert pred.size() == soft_label.size()
    if not detach_target:
        raise NotImplementedError('gradient into the soft label')
    rows = _RowLoss.apply(_launch_kl, pred, weight, soft_label, T)
    return _reduce_weighted(rows, reduction, avg_factor)


@LOSSES.register_module()
class KnowledgeDistillationKLDivLoss(nn.Module):
    """kd_loss.py:39-88."""

    def __init__(self, reduction='mean', loss_weight=1.0, T=10):
        super().__init__()
        assert T >= 1
        self.reduction, self.loss_weight, self.T = reduction, loss_weight, T

    def forward(self, pred, soft_label, weight=None, avg_factor=None,
                reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * knowledge_distillation_kl_div_loss(
            pred, soft_label, self.T, weight, reduction, avg_factor)


# The string the reference's side heads default to (ld_retina.py:32 etc.); it
# is unregistered there -- registering it as an alias is harmless and matches
# upstream mmdet >= 2.11.
LOSSES.register_module(name='LocalizationDistillationLoss',
                       module=type('LocalizationDistillationLoss',
                                   (KnowledgeDistillationKLDivLoss, ), {}))


@LOSSES.register_module()
class IMLoss(nn.Module):
    """kd_loss.py:91-120: loss_weight * mse(x, soft_target) (mean)."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, x, soft_target, weight=None, avg_factor=None,
                reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        # F.mse_loss mean-reduces before weighted_loss sees it, so every
        # reduction mode yields the same scalar (kd_loss.py:91-96).  As a KL
        # over K = rows*C entries it is not; a 1-row launch of the row kernel
        # is pointless, so use the fused block (LDHead) for the real path and
        # plain device arithmetic here.
        L.require_device(x, torch.float32, 'x')
        d = x - soft_target.detach()
        return self.loss_weight * (d * d).mean()


@LOSSES.register_module()
class QualityFocalLoss(nn.Module):
    """gfocal_loss.py:77-135."""

    def __init__(self, use_sigmoid=True, beta=2.0, reduction='mean',
                 loss_weight=1.0):
        super().__init__()
        self.use_sigmoid, self.beta = use_sigmoid, beta
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None,
                reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if not self.use_sigmoid or self.beta != 2.0:
            raise NotImplementedError('QFL with use_sigmoid=False / beta != 2')
        assert len(target) == 2
        label, score = target
        rows = _RowLoss.apply(_launch_qfl, pred, weight, label, score)
        return self.loss_weight * _reduce_weighted(rows, reduction, avg_factor)


@LOSSES.register_module()
class DistributionFocalLoss(nn.Module):
    """gfocal_loss.py:138-179."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None,
                reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        rows = _RowLoss.apply(_launch_dfl, pred, weight, target)
        return self.loss_weight * _reduce_weighted(rows, reduction, avg_factor)


@LOSSES.register_module()
class GIoULoss(nn.Module):
    """iou_loss.py:325-360."""

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None,
                reduction_override=None, **kwargs):
        # The reference returns (pred * weight).sum() = 0 early when no weight is
        # positive (iou_loss.py:346-348), which costs a host sync per call; the
        # row kernel yields the same zero (rows are multiplied by the weight)
        # without looking at the weights on the host.
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        rows = _RowLoss.apply(_launch_giou, pred, weight, target, self.eps)
        return self.loss_weight * _reduce_weighted(rows, reduction, avg_factor)


@LOSSES.register_module()
class FocalLoss(nn.Module):
    """losses/focal_loss.py:111-181 (sigmoid focal loss).  Evaluated inside the
    fused loss block of LDATSSHead (hp flag LD_LOSS_ATSS, gamma = 2); the
    module carries the hyper-parameters and is registrable for the configs."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25,
                 reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            'FocalLoss runs inside the fused LDATSSHead loss block '
            '(ld_loss_main_parts with LD_LOSS_ATSS); a stand-alone launch is '
            'not built')


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """losses/cross_entropy_loss.py:93-214; only the use_sigmoid=True form the
    ATSS centerness branch uses, evaluated by ld_loss_centerness."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean',
                 class_weight=None, loss_weight=1.0):
        super().__init__()
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.class_weight = reduction, class_weight
        self.loss_weight = loss_weight

    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            'CrossEntropyLoss (centerness) runs inside the fused LDATSSHead '
            'loss block (ld_loss_centerness)')


@LOSSES.register_module()
class IoULoss(nn.Module):
    """Only *registrable* (FCOSGFLHead's constructor default,
    fcos_gfl_head.py:111); every FCOS-GFL config overrides it with GIoULoss."""

    def __init__(self, linear=False, eps=1e-6, reduction='mean',
                 loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, *args, **kwargs):
        raise NotImplementedError('IoULoss is not on the LD train path')


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    """Only *registrable* (AnchorHead's constructor default, anchor_head.py:47-48,
    inherited by RetinaGFLHead); every RetinaGFL config overrides it with
    GIoULoss on decoded boxes."""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = (beta, reduction,
                                                       loss_weight)

    def forward(self, *args, **kwargs):
        raise NotImplementedError('SmoothL1Loss is not on the LD train path')


@LOSSES.register_module()
class CIoULoss(nn.Module):
    """Only *registrable*: the GFL teacher configs name it
    (configs/gfl/gfl_r50_fpn_1x_coco.py:43) but a frozen teacher never
    evaluates a loss."""

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, *args, **kwargs):
        raise NotImplementedError('CIoULoss is not on the LD train path')
