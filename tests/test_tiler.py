"""Row f3 (on-device tiling / normalising front end): the numpy oracle against the reference's own statements (CPU), the
HIP kernel against the oracle bit for bit (GPU), the batch it builds against the step's input contract."""
import numpy as np
import pytest
import torch


def test_oracle_blockshaped_matches_reference_contract():
    from oracle import tiler_oracle as to

    img = np.arange(1024 * 1024 * 3, dtype=np.uint32).reshape(1024, 1024, 3)
    g = to.blockshaped(img, 256, 256)
    assert g.shape == (16, 256, 256, 3)                         # the reference's own assertion (bcss.py:176)
    for k in range(16):                                         # row-major block order, physical layout preserved
        r, c = divmod(k, 4)
        assert np.array_equal(g[k], img[256 * r:256 * r + 256, 256 * c:256 * c + 256])
    # an exact-size box is a copy, Normalize is albumentations' fp32 order
    blk = (np.random.default_rng(0).integers(0, 256, (256, 256, 3))).astype(np.uint8)
    v = to.view(blk, 1, None, [(16, 16, 224, 224)], None, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    want = (blk[16:240, 16:240].astype(np.float32) - np.float32([0.485, 0.456, 0.406]) * np.float32(255)) \
        * np.reciprocal(np.float32([0.229, 0.224, 0.225]) * np.float32(255), dtype=np.float32)
    assert np.array_equal(v[0], want.transpose(2, 0, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("grid,hw", [(4, 1024), (1, 1024), (4, 512), (2, 96)])
def test_tile_views_match_oracle(hip_lib, grid, hw):
    from msf_wsi_amd import data, kernels as kn
    from oracle import tiler_oracle as to

    B, K = 3, grid * grid
    g = torch.Generator().manual_seed(grid * 100 + hw)
    img = torch.randint(0, 256, (B, hw, hw, 3), generator=g, dtype=torch.uint8)
    bs = hw // grid
    perm = torch.stack([torch.randperm(K, generator=g) for _ in range(B)])
    boxes = torch.tensor([[data.random_resized_crop_box(bs, bs, g) for _ in range(K)] for _ in range(B)],
                         dtype=torch.int32)
    boxes[0, 0] = torch.tensor([0, 0, bs, bs], dtype=torch.int32)             # whole block
    if bs >= 64:
        boxes[0, K - 1] = torch.tensor([bs - 64, bs - 64, 64, 64], dtype=torch.int32)  # exact-size box at the far corner
    flips = (torch.rand(B, K, generator=g) < 0.5).to(torch.uint8)
    size = 224 if bs >= 224 else 64
    if bs >= 64:
        boxes[1, 0] = torch.tensor([1, 2, size if size <= bs - 2 else bs - 2, size if size <= bs - 2 else bs - 2],
                                   dtype=torch.int32)
    out = kn.tile_views(img.cuda(), grid, perm.cuda(), boxes.cuda(), flips.cuda(), data.MEAN, data.STD, size)
    inv = kn.inverse_perm(perm.cuda())
    torch.cuda.synchronize()
    assert torch.equal(inv.cpu(), torch.argsort(perm, dim=1))               # jigsaw_reverse_idx (bcss.py:172)
    for b in range(B):
        ref = to.view(img[b].numpy(), grid, perm[b].numpy(), boxes[b].numpy(), flips[b].numpy(), data.MEAN, data.STD,
                      size)
        got = out[b].cpu().numpy()
        assert got.shape == ref.shape == (K, 3, size, size)
        assert np.array_equal(got, ref), float(np.abs(got - ref).max())


@pytest.mark.gpu
def test_device_tiler_builds_the_step_batch(hip_lib):
    """DeviceTiler.batch -> exactly the structure MSFWSI.forward / PretrainStep.step consume; un-shuffling the target
    tiles with the returned reverse index restores the spatial block order (what backbone.py:147-158 relies on)"""
    from msf_wsi_amd import data

    B = 2
    g = torch.Generator().manual_seed(9)
    tiles = [torch.randint(0, 256, (B, 1024, 1024, 3), generator=g, dtype=torch.uint8).cuda() for _ in range(2)]
    tiler = data.DeviceTiler(scale=4, crop_scale=(1.0, 1.0), flip_p=0.0)    # geometry-only: whole blocks, no flips
    (c1, c2), (t1, t2), idx = tiler.batch(tiles, tiles, gen=g)
    torch.cuda.synchronize()
    assert c1.shape == c2.shape == (B, 3, 224, 224) and t1.shape == t2.shape == (B * 16, 3, 224, 224)
    assert all(i.shape == (B, 16) and i.dtype == torch.int64 for i in idx)
    # target tile k of the batch is block perm[k]; gathering with the reverse index puts block j back at position j
    t = t1.view(B, 16, 3, 224, 224)
    restored = t[torch.arange(B).view(B, 1), idx[0]]
    plain = tiler.view(tiles[0], 4, None, torch.tensor([[[0, 0, 256, 256]] * 16] * B, dtype=torch.int32), None)
    assert torch.equal(restored, plain)
