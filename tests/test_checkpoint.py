"""Checkpoint wire format, CPU only.

The oracle is the layout mmcv.runner (1.2.4..1.3, un-vendored dependency of the
reference) publishes, the reference's own call sites (kd_one_stage.py:42-44,
resnet.py:597-599, tools/train.py:168-173), the state_dict key lists of the
REAL reference models (tests/golden/e2e.npz, produced by running the reference)
and torch.optim.SGD itself for the optimizer entry."""
import os
import warnings
from collections import OrderedDict

import numpy as np
import pytest
import torch

from ld_amd import checkpoint as CK


@pytest.fixture(scope='module')
def det():
    from ld_amd import model_zoo
    return model_zoo.build_seeded_ld_detector(18, 101, torch.device('cpu'))


def test_save_layout_and_reference_key_names(det, golden, tmp_path):
    f = str(tmp_path / 'work_dirs' / 'epoch_1.pth')
    meta = dict(mmdet_version='2.10.0+abcdef0', CLASSES=('a', 'b'), epoch=1,
                iter=7330)
    CK.save_checkpoint(det, f, meta=meta)
    ck = torch.load(f, map_location='cpu', weights_only=False)
    assert set(ck) == {'meta', 'state_dict'}
    assert ck['meta']['mmdet_version'] == meta['mmdet_version']
    assert ck['meta']['epoch'] == 1 and ck['meta']['iter'] == 7330
    assert 'mmcv_version' in ck['meta'] and 'time' in ck['meta']
    sd = ck['state_dict']
    assert isinstance(sd, OrderedDict)
    # exactly the reference student's keys and shapes, in its order; nothing of
    # the teacher (kd_one_stage.py:97-108)
    g = golden['e2e']
    assert list(sd.keys()) == [str(k) for k in g['tiny_r18_student_keys']]
    for k, shp in zip(sd.keys(), g['tiny_r18_student_shapes']):
        want = tuple(int(v) for v in str(shp).split('x')) if str(shp) else ()
        assert tuple(sd[k].shape) == want, k
    assert not any('teacher' in k for k in sd)
    assert all(v.device.type == 'cpu' for v in sd.values())


def test_roundtrip_and_module_prefix(det, tmp_path):
    from ld_amd import model_zoo
    f = str(tmp_path / 'a.pth')
    CK.save_checkpoint(det, f)
    ck = torch.load(f, map_location='cpu', weights_only=False)
    # a DDP-saved file carries 'module.' prefixes (mmcv strips them on load)
    ck['state_dict'] = OrderedDict(('module.' + k, v)
                                   for k, v in ck['state_dict'].items())
    f2 = str(tmp_path / 'ddp.pth')
    torch.save(ck, f2)
    from ld_amd.registry import build_detector
    other = build_detector(model_zoo.ld_detector(18, 101))
    with torch.no_grad():
        for p in other.parameters():
            p.add_(1.0)
    out = CK.load_checkpoint(other, f2)
    assert out['_load_report'] == dict(missing=[], unexpected=[], mismatched=[])
    for (k, a), (_, b) in zip(det.state_dict().items(),
                              other.state_dict().items()):
        assert torch.equal(a, b), k


def test_mismatch_reporting(det, tmp_path):
    sd = OrderedDict(det.state_dict())
    first = next(iter(sd))
    sd.pop(first)
    sd['fc.weight'] = torch.zeros(3)
    k2 = 'bbox_head.gfl_cls.weight'
    sd[k2] = torch.zeros(5, 5)
    f = str(tmp_path / 'bad.pth')
    torch.save(dict(state_dict=sd, meta={}), f)
    from ld_amd import model_zoo
    from ld_amd.registry import build_detector
    other = build_detector(model_zoo.ld_detector(18, 101))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        rep = CK.load_checkpoint(other, f)['_load_report']
    assert rep['missing'] == [first]
    assert rep['unexpected'] == ['fc.weight']
    assert [m[0] for m in rep['mismatched']] == [k2]
    assert any('do not match exactly' in str(x.message) for x in w)
    with pytest.raises(RuntimeError):
        CK.load_checkpoint(other, f, strict=True)


def test_torchvision_scheme_resolves_offline(tmp_path, monkeypatch):
    """configs/ld/*.py: pretrained='torchvision://resnet18'.  The zoo file is a
    BARE state_dict with torchvision key names (= mmdet's ResNet key names);
    fc.* is unexpected, nothing is missing (resnet.py:597-599)."""
    from ld_amd.registry import BACKBONES, build_from_cfg
    cfg = dict(type='ResNet', depth=18, num_stages=4, out_indices=(0, 1, 2, 3),
               frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=True),
               norm_eval=True, style='pytorch')
    src = build_from_cfg(cfg, BACKBONES)
    src.init_weights(None)
    zoo = OrderedDict((k, v.clone() + 0.25) for k, v in src.state_dict().items())
    zoo['fc.weight'] = torch.zeros(1000, 512)
    zoo['fc.bias'] = torch.zeros(1000)
    d = tmp_path / 'hub' / 'checkpoints'
    d.mkdir(parents=True)
    torch.save(zoo, str(d / 'resnet18-5c106cde.pth'))
    monkeypatch.setenv('TORCH_HOME', str(tmp_path))
    monkeypatch.delenv('LD_CHECKPOINT_DIR', raising=False)
    assert CK.resolve_checkpoint_path('torchvision://resnet18') == \
        str(d / 'resnet18-5c106cde.pth')
    dst = build_from_cfg(cfg, BACKBONES)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        dst.init_weights('torchvision://resnet18')
    for k, v in dst.state_dict().items():
        if 'num_batches_tracked' in k:
            continue
        assert torch.equal(v, zoo[k]), k
    with pytest.raises(FileNotFoundError) as e:
        CK.resolve_checkpoint_path('torchvision://resnet50')
    assert 'resnet50-19c8e357.pth' in str(e.value)
    with pytest.raises(FileNotFoundError) as e:
        CK.resolve_checkpoint_path(
            'https://download.openmmlab.com/mmdetection/v2.0/gfl/'
            'gfl_r101_fpn_mstrain_2x_coco/gfl_r101_fpn_mstrain_2x_coco_'
            '20200629_200126-dd12f847.pth')
    assert 'gfl_r101_fpn_mstrain_2x_coco_20200629_200126-dd12f847.pth' in \
        str(e.value)


def test_optimizer_state_is_torch_sgd_wire_format(det, tmp_path):
    """The flat momentum arena <-> torch.optim.SGD.state_dict(), both ways,
    through a checkpoint file."""
    from ld_amd.train import SGDTrainer
    tr = SGDTrainer(det, lr=0.01, momentum=0.9, weight_decay=1e-4)
    allp = list(det.parameters())
    trainable = [p for p in allp if p.requires_grad]
    assert 0 < len(trainable) < len(allp)  # frozen stem / stage 1
    gen = torch.Generator().manual_seed(3)
    tr.flat_momentum.copy_(torch.randn(tr.flat_momentum.shape, generator=gen))
    tr.iter = 5
    f = str(tmp_path / 'iter_5.pth')
    CK.save_checkpoint(det, f, optimizer=tr, meta=dict(epoch=0, iter=5))
    ck = torch.load(f, map_location='cpu', weights_only=False)
    osd = ck['optimizer']
    # torch's own optimizer accepts it ...
    ref = torch.optim.SGD(allp, lr=0.5, momentum=0.1, weight_decay=0.0)
    ref.load_state_dict(osd)
    g = ref.param_groups[0]
    assert (g['lr'], g['momentum'], g['weight_decay']) == (0.01, 0.9, 1e-4)
    assert g['nesterov'] is False and g['dampening'] == 0
    assert set(ref.state_dict()['state']) == \
        {i for i, p in enumerate(allp) if p.requires_grad}
    for p, o in zip(tr.arena.order, tr.arena.offsets):
        buf = ref.state[p]['momentum_buffer']
        assert torch.equal(buf.reshape(-1),
                           tr.flat_momentum[o:o + p.numel()]), 'momentum'
    # ... and what torch's optimizer writes loads back into the arena
    for st in ref.state.values():
        st['momentum_buffer'].mul_(2.0)
    tr2 = SGDTrainer(det, lr=0.3)
    out = CK.resume(tr2, f)
    assert tr2.iter == 5 and tr2.epoch == 0 and out['meta']['iter'] == 5
    # (the arena pads every parameter to 64 floats; the pads belong to nobody)
    keep = torch.zeros_like(tr.flat_momentum, dtype=torch.bool)
    for p, o in zip(tr.arena.order, tr.arena.offsets):
        keep[o:o + p.numel()] = True
    assert torch.equal(tr2.flat_momentum[keep], tr.flat_momentum[keep])
    assert not bool(tr2.flat_momentum[~keep].any())
    assert (tr2.lr, tr2.momentum, tr2.weight_decay) == (0.01, 0.9, 1e-4)
    tr2.load_state_dict(ref.state_dict())
    assert torch.equal(tr2.flat_momentum[keep], 2.0 * tr.flat_momentum[keep])
    # a fresh trainer has taken no step: no momentum entries, like torch
    tr3 = SGDTrainer(det, lr=0.01)
    assert tr3.state_dict()['state'] == {}
    assert torch.optim.SGD(allp, lr=0.01, momentum=0.9).state_dict()['state'] \
        == {}


def test_teacher_checkpoint_call_site(tmp_path, monkeypatch):
    """kd_one_stage.py:42-44: the teacher is filled from teacher_ckpt, is not
    part of the student's parameters / state_dict, and a missing file is
    reported."""
    from ld_amd import model_zoo
    from ld_amd.registry import build_detector
    t_src = build_detector(model_zoo.gfl_detector(18))
    with torch.no_grad():
        for p in t_src.parameters():
            p.fill_(0.125)
    f = str(tmp_path / 'gfl_teacher.pth')
    CK.save_checkpoint(t_src, f, meta=dict(CLASSES=('x', )))
    cfg = model_zoo.ld_detector(18, 18)
    cfg['teacher_ckpt'] = f
    det = build_detector(cfg)
    tw = next(det.teacher_model.parameters())
    assert float(tw.flatten()[0]) == 0.125
    assert not any(k.startswith('teacher') for k in det.state_dict())
    n_student = sum(1 for _ in det.parameters())
    assert n_student == sum(1 for _ in t_src.parameters())
    # a missing teacher checkpoint raises, as mmcv's load_checkpoint does
    # (a random teacher would distil finite, meaningless losses) ...
    cfg = model_zoo.ld_detector(18, 18)
    cfg['teacher_ckpt'] = str(tmp_path / 'nope.pth')
    monkeypatch.delenv('LD_ALLOW_MISSING_CKPT', raising=False)
    with pytest.raises(FileNotFoundError):
        build_detector(cfg)
    cfg['pretrained'] = 'torchvision://resnet18'
    cfg['teacher_ckpt'] = None
    monkeypatch.setenv('TORCH_HOME', str(tmp_path / 'empty_hub'))
    monkeypatch.delenv('LD_CHECKPOINT_DIR', raising=False)
    with pytest.raises(FileNotFoundError):
        build_detector(cfg)
    # ... unless the caller opts in (synthetic-throughput runs)
    monkeypatch.setenv('LD_ALLOW_MISSING_CKPT', '1')
    cfg['teacher_ckpt'] = str(tmp_path / 'nope.pth')
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        build_detector(cfg)
    assert any('teacher_ckpt not loaded' in str(x.message) for x in w)
    assert any('backbone randomly' in str(x.message) for x in w)
