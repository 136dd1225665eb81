"""CPU: host logic of block.py that needs no device -- the light-shape rules of the many-lights entries and the signature that
ties a march call to its prepass (round 6)."""
import pytest
import torch


def test_light_shapes_one_light_and_many_lights_forms():
    from geomconsistentfr_amd import block as R
    from geomconsistentfr_amd._lib import GcfrError
    B = 4
    l3, a2, L, multi = R._light_shapes(B, torch.zeros(B, 3), torch.zeros(B))
    assert (tuple(l3.shape), tuple(a2.shape), L, multi) == ((B, 1, 3), (B, 1), 1, False)
    l3, a2, L, multi = R._light_shapes(B, torch.zeros(B, 3, 1, 1), torch.zeros(B, 1, 1))          # the scripts' target_lighting (S1:588)
    assert (tuple(l3.shape), L, multi) == ((B, 1, 3), 1, False)
    l3, a2, L, multi = R._light_shapes(B, torch.zeros(B, 11, 3), torch.zeros(B, 11))
    assert (tuple(l3.shape), tuple(a2.shape), L, multi) == ((B, 11, 3), (B, 11), 11, True)
    l3, a2, L, multi = R._light_shapes(B, torch.zeros(B, 1, 3), torch.zeros(B, 1))                # L = 1 WITH a light axis
    assert (L, multi) == (1, True)
    sl = torch.zeros(B, 1, 1, 4)
    l3, a2, L, multi = R._light_shapes(B, sl[:, 0, 0, 1:4], sl[:, 0, 0, 0])                        # the training form's slices (T8:357, 367)
    assert l3.data_ptr() == sl[:, 0, 0, 1:4].data_ptr() and not multi                             # a view: no copy
    with pytest.raises(GcfrError):
        R._light_shapes(B, torch.zeros(B, 11, 3), torch.zeros(B, 10))
    with pytest.raises(GcfrError):
        R._light_shapes(B, torch.zeros(B, 4), torch.zeros(B))


def test_source_signature_sees_other_tensors_views_and_in_place_writes():
    from geomconsistentfr_amd import block as R
    d, m, l = torch.rand(2, 1, 8, 8), torch.ones(2, 8, 8), torch.rand(2, 3)
    s0 = R.source_signature(d, m, l)
    assert s0 == R.source_signature(d, m, l)
    assert s0 == R.source_signature(d.detach(), m, l.detach())                # autograd's detached aliases: same storage, same counter
    assert s0 != R.source_signature(d.clone(), m, l)                          # another tensor with the same values
    assert s0 != R.source_signature(d, m, l[:, :3].flip(0))                   # another view / layout
    assert s0 != R.source_signature(d.reshape(2, 8, 8), m, l)                 # the same storage under another shape
    d.add_(1.0)                                                               # written in place: the version counter moves ...
    assert s0 != R.source_signature(d, m, l)
    v = d.reshape(2, 8, 8)
    s1 = R.source_signature(v, m, l)
    d.mul_(2.0)                                                               # ... for every view of the tensor
    assert s1 != R.source_signature(v, m, l)


def test_normals_stage_rule_and_result_shapes():
    from geomconsistentfr_amd import block as R
    assert R.normals_stage_for(1) == "fused" and R.normals_stage_for(11) == "fused" and R.normals_stage_for(18) == "kernel"
    B, L, H, W = 2, 3, 4, 5
    z = lambda *s: torch.zeros(*s)
    r = R._result_dict(B, H, W, True, z(B, L), z(B, L, H, W), z(B, L, H, W), z(B, L, H, W), z(B, L, 3, H, W), z(B, L, 3), z(B, L, H, W),
                       normals=z(B, 3, H, W))
    assert tuple(r["rendered_images"].shape) == (B, L, 3, H, W) and tuple(r["unit_light_direction"].shape) == (B, L, 3, 1, 1)
    assert tuple(r["ambient_light"].shape) == (B, L, H, W) and tuple(r["ambient_values"].shape) == (B, L, 1, 1)
    r = R._result_dict(B, H, W, False, z(B, 1), z(B, 1, H, W), z(B, 1, H, W), z(B, 1, H, W), z(B, 1, 3, H, W), z(B, 1, 3), z(B, 1, H, W))
    assert tuple(r["rendered_images"].shape) == (B, 3, H, W) and tuple(r["unit_light_direction"].shape) == (B, 3, 1, 1)   # T8:524
    assert tuple(r["ambient_light"].shape) == (B, H, W) and tuple(r["ambient_values"].shape) == (B, 1, 1)


def test_camera_scalars_cache_follows_the_tensor_object_and_its_version():
    """camera_scalars caches per tensor OBJECT (host tensors too since round 6: the read is 10 us of tensor arithmetic per call):
    an in-place write invalidates the entry, a dead tensor's entry is dropped, per-image matrices give None."""
    import gc
    from geomconsistentfr_amd import block as R
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = 1570.0
    K[:, 2, 2] = 1.0
    K[:, 0, 2], K[:, 1, 2] = 128.0, 120.0
    n0 = len(R._CAMERA_CACHE)
    assert R.camera_scalars(K) == (1570.0, 1570.0, 128.0, 120.0) and len(R._CAMERA_CACHE) == n0 + 1
    assert R.camera_scalars(K) == (1570.0, 1570.0, 128.0, 120.0)                     # served from the cache
    K[:, 0, 0] = 700.0                                                                # in place: the version counter moves
    assert R.camera_scalars(K)[0] == 700.0
    two = torch.cat([K, K]).clone()
    assert R.camera_scalars(two) == (700.0, 1570.0, 128.0, 120.0)                     # B equal matrices
    two[1, 0, 0] = 900.0
    assert R.camera_scalars(two) is None                                              # per-image matrices: the three-stage path
    del K, two
    gc.collect()
    assert len(R._CAMERA_CACHE) == n0
