"""CPU: the hipGraph step's handling of the batch dicts the reference's collate produces (data/datasets/kitti_v2.py:L196-221): tensors,
lists of numpy arrays (ctx_img / ctx_img_orig), one python bool `flip` for the whole batch, `metadata` (list of dicts).  The host logic
(static-buffer construction, per-batch copy, graph-set key) is exercised without a GPU; tests/test_gpu_models.py runs it under real replays."""
import numpy as np
import pytest
import torch

from simpledepthestimation_amd.engine.trainer import HipTrainer, ParamGroup
from test_dp_gloo import Tiny, torch_adam


def _collated(seed, flip):
    g = np.random.default_rng(seed)
    B, H, W = 2, 4, 6
    return {"img": torch.from_numpy(g.random((B, 3, H, W), dtype=np.float32)), "img_orig": torch.from_numpy(g.random((B, 3, H, W), dtype=np.float32)),
            "ctx_img": [g.random((B, 3, H, W), dtype=np.float32), g.random((B, 3, H, W), dtype=np.float32)],
            "ctx_img_orig": [g.random((B, 3, H, W), dtype=np.float32), g.random((B, 3, H, W), dtype=np.float32)],
            "intrinsics": torch.eye(3).repeat(B, 1, 1), "flip": flip,
            "metadata": [{"date": "2011_09_26", "drive": "0001", "img_id": f"{seed:010d}", "h_before_resize": 375}, {"date": "2011_09_26", "drive": "0002", "img_id": "7"}]}


def _trainer():
    torch.manual_seed(0)
    m = Tiny()
    return HipTrainer(m, [ParamGroup("all", m.named_parameters(), 1e-2, 0.0)], adam_fn=torch_adam)


def test_static_batch_takes_numpy_lists_flip_and_metadata():
    tr = _trainer()
    b0, b1 = _collated(0, False), _collated(1, True)
    tr._static_batch = tr._to_static(b0)
    s = tr._static_batch
    assert torch.is_tensor(s["img"]) and s["img"].data_ptr() != b0["img"].data_ptr()            # cloned: static address of its own
    assert isinstance(s["ctx_img"], list) and all(torch.is_tensor(t) for t in s["ctx_img"])     # numpy -> tensors
    assert s["metadata"] is b0["metadata"] and s["flip"] is False
    ptrs = [s["img"].data_ptr()] + [t.data_ptr() for t in s["ctx_img"] + s["ctx_img_orig"]]
    tr._copy_into_static(b1)                                                                   # no exception for flip / metadata / numpy
    s = tr._static_batch
    assert ptrs == [s["img"].data_ptr()] + [t.data_ptr() for t in s["ctx_img"] + s["ctx_img_orig"]], "static tensors must keep their addresses"
    assert torch.equal(s["img"], b1["img"]) and np.array_equal(s["ctx_img"][1].numpy(), b1["ctx_img"][1]) and np.array_equal(s["ctx_img_orig"][0].numpy(), b1["ctx_img_orig"][0])
    assert s["flip"] is True and s["metadata"] is b1["metadata"]
    # the graph set is keyed by the steering scalars only
    assert tr._graph_key(b0) != tr._graph_key(b1) and tr._graph_key(b0) == tr._graph_key(_collated(5, False))
    assert tr._graph_key(b1) == (("flip", True),)
    with pytest.raises(RuntimeError):
        tr._copy_into_static({k: v for k, v in b0.items() if k != "intrinsics"})
    bad = _collated(2, False); bad["ctx_img"] = bad["ctx_img"][:1]
    with pytest.raises(RuntimeError):
        tr._copy_into_static(bad)


def test_set_lr_is_host_side_and_reaches_the_optimizer():
    seen = []

    def spy(p, g, m, v, seg_end, seg_lr, seg_wd, *a, **k):
        seen.append((list(seg_end), list(seg_lr), list(seg_wd)))
        return torch_adam(p, g, m, v, seg_end, seg_lr, seg_wd, *a, **k)
    torch.manual_seed(0)
    m = Tiny()
    tr = HipTrainer(m, [ParamGroup("a", m.a.named_parameters(prefix="a"), 1e-2, 1e-2), ParamGroup("b", m.b.named_parameters(prefix="b"), 5e-3, 0.0)], adam_fn=spy)
    g = torch.Generator().manual_seed(1)
    batch = {"x": torch.randn(8, 6, generator=g), "t": torch.randn(8, generator=g)}
    tr.step(batch)
    tr.set_lr([3e-3, 4e-3])
    tr.step(batch)
    assert seen[0] == ([40, 52], [1e-2, 5e-3], [1e-2, 0.0]) and seen[1][1] == [3e-3, 4e-3]      # 30 + 5 -> 32 + 8; 5 + 1 -> 8 + 4 (16-byte slots)
    with pytest.raises(ValueError):
        tr.set_lr([1.0])
