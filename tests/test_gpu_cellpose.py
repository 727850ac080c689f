"""GPU tests of the Cellpose path (BASELINE configs[4]): the HIP flow -> mask post-processing against the CPU
restatement of the published algorithm (oracle/cellpose_dynamics.py -- PARITY UNPINNED: cellpose itself is not
available offline and the reference's tests mock it, RT/test_model.py:124-376), and the SegmentationModel surface
around it (R/model.py:171-290: int64 labels, per-image failure -> SegmentationWarning + None)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cellpose_masks_vs_oracle():
    from arcadia_microscopy_tools_amd import hipops
    from arcadia_microscopy_tools_amd.device import get_context
    from arcadia_microscopy_tools_amd import synth
    from oracle import cellpose_dynamics as cd

    ctx = get_context()
    cases = [((96, 120), 8, 0, 0.0, 60), ((160, 200), 18, 3, 0.0, 200), ((130, 97), 10, 5, 0.3, 100),
             ((64, 64), 0, 1, 0.0, 20)]
    for shape, ncells, seed, noise, niter in cases:
        dP, prob, truth = synth.synthetic_flows(shape, ncells, seed=seed, noise=noise)
        ref = cd.compute_masks(dP, prob, niter=niter)
        lab, cnt = hipops.cellpose_masks(ctx.asarray(dP[None]), ctx.asarray(prob[None]), niter=niter)
        got = lab.numpy()[0]
        assert int(cnt.numpy()[0]) == int(ref.max()), (shape, seed)
        assert np.array_equal(got, ref), (shape, seed, int((got != ref).sum()))
        if ncells and noise == 0.0:  # every synthetic cell comes back as (mostly) one label
            for k in range(1, int(truth.max()) + 1):
                vals = got[truth == k]
                if vals.size:
                    assert np.bincount(vals).argmax() > 0
    # thresholds and size filters
    dP, prob, _ = synth.synthetic_flows((120, 150), 12, seed=9)
    for thr, min_size, frac in ((0.0, 15, 0.4), (7.0, 15, 0.4), (0.0, 400, 0.4), (0.0, 15, 0.01)):
        ref = cd.compute_masks(dP, prob, cellprob_threshold=thr, niter=80, min_size=min_size, max_size_fraction=frac)
        lab, cnt = hipops.cellpose_masks(ctx.asarray(dP[None]), ctx.asarray(prob[None]), cellprob_threshold=thr, niter=80,
                                         min_size=min_size, max_size_fraction=frac)
        assert np.array_equal(lab.numpy()[0], ref) and int(cnt.numpy()[0]) == int(ref.max())
    # a batch of planes in one call, and the seed-capacity flag
    dP2 = np.stack([synth.synthetic_flows((80, 96), 6, seed=s)[0] for s in (1, 2)])
    pr2 = np.stack([synth.synthetic_flows((80, 96), 6, seed=s)[1] for s in (1, 2)])
    lab, cnt = hipops.cellpose_masks(ctx.asarray(dP2), ctx.asarray(pr2), niter=60)
    for b in range(2):
        assert np.array_equal(lab.numpy()[b], cd.compute_masks(dP2[b], pr2[b], niter=60))
    _, cnt = hipops.cellpose_masks(ctx.asarray(dP2), ctx.asarray(pr2), niter=60, max_seeds=2)
    assert (cnt.numpy() == -1).all()


def test_segmentation_model_network_backend():
    """SegmentationModel(backend='cellpose-hip'): a flow network (here a module that returns precomputed flows) in bf16
    on the GPU + the HIP post-processing; segment() returns int64 labels equal to the oracle on the same flows,
    batch_segment() turns a failing image into a SegmentationWarning and None (R/model.py:276-288)."""
    import torch

    from arcadia_microscopy_tools_amd.exceptions import SegmentationWarning
    from arcadia_microscopy_tools_amd.model import SegmentationModel
    from arcadia_microscopy_tools_amd import synth
    from oracle import cellpose_dynamics as cd

    H, W = 112, 144
    dP, prob, _ = synth.synthetic_flows((H, W), 9, seed=4)
    flows = torch.from_numpy(np.concatenate([dP, prob[None]])[None])

    class Fixed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("flows", flows)

        def forward(self, x):
            if x.shape[-2:] != self.flows.shape[-2:]:
                raise ValueError("unexpected image size")
            return self.flows.to(x.dtype).expand(x.shape[0], -1, -1, -1)

    model = SegmentationModel(backend="cellpose-hip", network=Fixed(), compute_dtype="fp32")
    img = np.random.default_rng(0).random((2, H, W))
    out = model.segment(img, num_iterations=100)
    assert out.dtype == np.int64 and out.shape == (H, W)
    assert np.array_equal(out, cd.compute_masks(dP, prob, niter=100))
    with pytest.warns(SegmentationWarning, match="failed on image 1"):
        res = model.batch_segment([img, np.zeros((2, 32, 32)), img[0]], num_iterations=100, show_progress=False)
    assert res[1] is None and np.array_equal(res[0], out) and np.array_equal(res[2], out)
    with pytest.raises(ValueError, match="between -10 and 10"):
        model.segment(img, cellprob_threshold=11)
    with pytest.raises(ValueError, match="needs network="):
        SegmentationModel(backend="cellpose-hip")


def test_standin_forward_bf16():
    """The random-weight CPnet stand-in runs in bf16 channels-last through PyTorch-ROCm and its (meaningless) output
    goes through the post-processing without error; only shapes / dtypes / finiteness are asserted."""
    import torch

    from arcadia_microscopy_tools_amd import cellpose_hip as ch

    dev = torch.device("cuda", 0)
    net, dt = ch.prepare_network(ch.make_standin(), dev, "bf16")
    x = torch.randn(2, 2, 256, 256, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = net(x)
    assert y.shape == (2, 3, 256, 256) and y.dtype == torch.bfloat16 and bool(torch.isfinite(y.float()).all())
    flops = ch.forward_flops(net, x)
    assert 7.0e10 < flops < 9.5e10  # ~40.5 GFLOP per 256^2 tile (648 GFLOP per 1024^2)
    labels, counts = ch.flows_to_masks(y.float(), niter=10, max_seeds=65536)
    assert labels.shape == (2, 256, 256) and counts.numpy().shape == (2,)


def test_flow_error_and_filter_vs_oracle():
    """The flow-error filter of compute_masks (remove_bad_flow_masks -> metrics.flow_error -> masks_to_flows) on the
    device against its CPU restatement (PARITY UNPINNED, see the module docstring): per-mask errors within 1e-9
    relative (float64 diffusion, sums in a different order), the filtered + renumbered label images identical.  The
    cases cover the three diffusion classes (LDS tile of 2,304 px, of 6,400 px, global planes) and every threshold
    between two masks' errors."""
    from arcadia_microscopy_tools_amd import hipops, synth
    from arcadia_microscopy_tools_amd.device import get_context
    from oracle import cellpose_dynamics as cd

    ctx = get_context()
    for shape, ncells, seed, noise, niter in (((96, 120), 8, 0, 0.0, 60), ((130, 97), 10, 5, 0.6, 100),
                                              ((160, 200), 18, 3, 1.0, 120)):
        dP, prob, _ = synth.synthetic_flows(shape, ncells, seed=seed, noise=noise)
        masks = cd.compute_masks(dP, prob, niter=niter, min_size=0)
        K = int(masks.max())
        ref = cd.flow_error(masks, dP)
        got = hipops.cellpose_flow_error(ctx.asarray(masks[None]), ctx.asarray(dP[None]), K).numpy()[0]
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12, err_msg=str((shape, seed)))
        # thresholds between consecutive errors: every possible verdict of the filter
        srt = np.sort(ref)
        for thr in [float(srt[0] / 2)] + [float((a + b) / 2) for a, b in zip(srt[:-1], srt[1:]) if b > a * (1 + 1e-6)][:6]:
            want = cd.compute_masks(dP, prob, niter=niter, flow_threshold=thr, min_size=15)
            lab, cnt = hipops.cellpose_masks(ctx.asarray(dP[None]), ctx.asarray(prob[None]), niter=niter, flow_threshold=thr)
            assert int(cnt.numpy()[0]) == int(want.max()), (shape, thr)
            assert np.array_equal(lab.numpy()[0], want), (shape, thr)
    # large masks: a 70 x 80 blob (LDS class 2) and a 150 x 170 blob (global planes) next to small ones
    yy, xx = np.mgrid[0:260, 0:300]
    masks = np.zeros((260, 300), np.int32)
    masks[(yy - 60) ** 2 / 35.0 ** 2 + (xx - 70) ** 2 / 40.0 ** 2 < 1] = 1
    masks[(yy - 150) ** 2 / 75.0 ** 2 + (xx - 200) ** 2 / 85.0 ** 2 < 1] = 2
    masks[(yy - 30) ** 2 + (xx - 250) ** 2 < 12 ** 2] = 3
    masks[(masks == 0) & ((yy - 230) ** 2 + (xx - 40) ** 2 < 15 ** 2)] = 4
    rng = np.random.default_rng(3)
    dP = rng.normal(0, 2.0, (2,) + masks.shape).astype(np.float32)
    ref = cd.flow_error(masks, dP)
    got = hipops.cellpose_flow_error(ctx.asarray(masks[None]), ctx.asarray(dP[None]), 4).numpy()[0]
    np.testing.assert_allclose(got, ref, rtol=1e-9)
    # the flows re-derived from the masks reproduce themselves: error ~ 0 when the network's flows are 5 x those
    mu, _ = cd.masks_to_flows(masks)
    got0 = hipops.cellpose_flow_error(ctx.asarray(masks[None]), ctx.asarray((5 * mu).astype(np.float32)[None]), 4).numpy()[0]
    assert (got0 < 1e-10).all()
    # a batch: two planes, different label counts
    m2 = np.stack([masks, np.where(masks == 2, 0, masks)])
    d2 = np.stack([dP, dP[::-1].copy()])
    got = hipops.cellpose_flow_error(ctx.asarray(m2), ctx.asarray(d2), 4).numpy()
    np.testing.assert_allclose(got[0], ref, rtol=1e-9)
    np.testing.assert_allclose(got[1], cd.flow_error(m2[1], d2[1]), rtol=1e-9)


def test_fill_holes_and_remove_small_masks_vs_oracle():
    """utils.fill_holes_and_remove_small_masks on the device against the CPU restatement (scipy binary_fill_holes per
    label, in sequence): rings, a mask nested in another's hole (the later / earlier label cases, which take the
    sequential kernel), overlapping holes, labels with gaps, the size floor, boxes beyond the LDS map."""
    from arcadia_microscopy_tools_amd import hipops
    from arcadia_microscopy_tools_amd.device import get_context
    from oracle import cellpose_dynamics as cd

    ctx = get_context()
    yy, xx = np.mgrid[0:120, 0:140]

    def ring(cy, cx, r0, r1):
        d = (yy - cy) ** 2 + (xx - cx) ** 2
        return (d >= r0 * r0) & (d < r1 * r1)

    cases = []
    a = np.zeros((120, 140), np.int32)
    a[ring(30, 30, 8, 14)] = 1                       # plain ring: hole filled
    a[ring(30, 80, 0, 9)] = 2                        # disk
    a[ring(80, 40, 10, 18)] = 5                      # ring with a gap in the numbering before it
    a[80, 40] = 0
    a[ring(85, 105, 0, 3)] = 6                       # below min_size 15? 3^2 * pi ~ 28 px: kept at 15, dropped at 40
    cases.append(("rings", a))
    b = a.copy()
    b[ring(80, 40, 0, 4)] = 7                        # a LATER label inside label 5's hole: overwritten, vanishes
    cases.append(("nested-later", b))
    c = a.copy()
    c[ring(80, 40, 0, 4)] = 3                        # an EARLIER label inside label 5's hole: keeps its number, loses its pixels
    cases.append(("nested-earlier", c))
    d = np.zeros((120, 140), np.int32)               # two C-shapes whose boxes (and enclosed regions) interleave
    d[20:60, 20:24] = 1; d[20:24, 20:70] = 1; d[56:60, 20:70] = 1; d[20:60, 66:70] = 1   # closed frame 1
    d[30:50, 30:34] = 2; d[30:34, 30:60] = 2; d[46:50, 30:60] = 2; d[30:50, 56:60] = 2   # closed frame 2 inside frame 1
    cases.append(("frames", d))
    e = np.zeros((300, 320), np.int32)               # a box of 250 x 270 = 67,500 px: beyond the 48 KB LDS map
    yy2, xx2 = np.mgrid[0:300, 0:320]
    dd = (yy2 - 150) ** 2 / 125.0 ** 2 + (xx2 - 160) ** 2 / 135.0 ** 2
    e[(dd < 1) & (dd > 0.8)] = 1
    e[(yy2 - 150) ** 2 + (xx2 - 160) ** 2 < 100] = 2
    cases.append(("big", e))
    rng = np.random.default_rng(11)
    f = (rng.random((90, 110)) < 0.55) * rng.integers(1, 9, (90, 110))
    cases.append(("noise", f.astype(np.int32)))      # cracked masks full of one-pixel holes holding other labels
    for name, lab in cases:
        K = int(lab.max())
        for min_size in (15, 40, 0):
            for fh in (True, False):
                want = cd.fill_holes_and_remove_small_masks(lab, min_size, fh)
                got, cnt = hipops.fill_holes_remove_small(ctx.asarray(lab[None]), K, min_size, fh)
                assert np.array_equal(got.numpy()[0], want), (name, min_size, fh, int((got.numpy()[0] != want).sum()))
                assert int(cnt.numpy()[0]) == int(want.max()) or name in ("nested-earlier", "noise"), (name, min_size, fh)
    # two planes at once: one nested (sequential kernel), one not
    both = np.stack([a, b])
    got, cnt = hipops.fill_holes_remove_small(ctx.asarray(both), int(both.max()), 15, True)
    for i in range(2):
        assert np.array_equal(got.numpy()[i], cd.fill_holes_and_remove_small_masks(both[i], 15, True)), i


def test_network_route_uses_every_parameter():
    """R/model.py:171-215 hands diameter, flow_threshold, cellprob_threshold, niter and batch_size to eval: on the
    network + HIP route each one acts (none is accepted and ignored), unknown eval options are refused, and the
    classical backend refuses Cellpose-only parameters."""
    import torch

    from arcadia_microscopy_tools_amd import cellpose_hip as ch, synth
    from arcadia_microscopy_tools_amd.model import SegmentationModel
    from oracle import cellpose_dynamics as cd

    H, W = 160, 208
    dP, prob, _ = synth.synthetic_flows((H, W), 14, seed=6, noise=0.8)
    flows = np.concatenate([dP, prob[None]]).astype(np.float32)

    class Lookup(torch.nn.Module):
        """A "network" whose output is a function of its input alone: channel 0 of the image carries the pixel index
        of a stored flow field, so tiles and resized images map to the matching crops."""

        def __init__(self):
            super().__init__()
            self.register_buffer("flows", torch.from_numpy(flows).reshape(3, -1))
            self.calls = []

        def forward(self, x):
            self.calls.append(tuple(x.shape))
            idx = x[:, 0].float().round().long().clamp(0, self.flows.shape[1] - 1)
            return self.flows[:, idx].permute(1, 0, 2, 3).to(x.dtype)

    index_img = np.arange(H * W, dtype=np.float32).reshape(1, H, W)
    net = Lookup()
    model = SegmentationModel(backend="cellpose-hip", network=net, compute_dtype="fp32")
    # whole image in one pass, no filter: the round-2 behaviour
    base = model.segment(index_img, flow_threshold=0, num_iterations=100, bsize=0, fill_holes=False)
    assert np.array_equal(base, cd.compute_masks(dP, prob, niter=100))
    # flow_threshold acts
    err = cd.flow_error(cd.compute_masks(dP, prob, niter=100, min_size=0), dP)
    thr = float(np.sort(err)[len(err) // 2] * 1.0001)
    filt = model.segment(index_img, flow_threshold=thr, num_iterations=100, bsize=0)
    want = cd.compute_masks(dP, prob, niter=100, flow_threshold=thr, fill_holes=True)
    assert np.array_equal(filt, want) and filt.max() < base.max()
    # batch_size acts: 256-px tiles with 10 % overlap over 160 x 208 -> one tile; 64-px tiles -> batches of batch_size
    net.calls.clear()
    tiled = model.segment(index_img, flow_threshold=0, num_iterations=100, bsize=64, batch_size=5, fill_holes=False)
    ys, _ = ch.tile_starts(H, 64)
    xs, _ = ch.tile_starts(W, 64)
    ntiles = len(ys) * len(xs)
    assert [c[0] for c in net.calls] == [5] * (ntiles // 5) + ([ntiles % 5] if ntiles % 5 else [])
    assert all(c[2:] == (64, 64) for c in net.calls)
    # a lookup network gives every tile the exact flows of its crop, so the taper-blended field is the field itself up
    # to float32 rounding; the masks are those of the oracle's blend of the same tiles (same float32 operations)
    ref_blend = cd.tiled_apply(lambda t: np.stack([flows.reshape(3, -1)[:, np.rint(tt[0]).astype(int)] for tt in t]),
                               index_img, bsize=64)
    np.testing.assert_allclose(ref_blend, flows, rtol=2e-6, atol=2e-6)
    assert np.array_equal(tiled, cd.compute_masks(ref_blend[:2], ref_blend[2], niter=100))
    # diameter acts: 60 px -> the image is halved before the network and the flows doubled back (bilinear)
    net.calls.clear()
    model.segment(index_img, cell_diameter_px=60, flow_threshold=0, num_iterations=20, bsize=0, fill_holes=False)
    assert net.calls == [(1, 1, 80, 112)]  # 80 x 104 padded to a multiple of 16
    x = torch.from_numpy(index_img).cuda()
    small = torch.nn.functional.interpolate(x[None], size=(80, 104), mode="bilinear", align_corners=False)[0]
    np.testing.assert_allclose(small.cpu().numpy(), cd.resize_bilinear(index_img, 80, 104), rtol=1e-6)
    back = torch.nn.functional.interpolate(small[None], size=(H, W), mode="bilinear", align_corners=False)[0]
    np.testing.assert_allclose(back.cpu().numpy(), cd.resize_bilinear(cd.resize_bilinear(index_img, 80, 104), H, W), rtol=1e-6)
    # nothing is silently dropped
    with pytest.raises(RuntimeError, match="does not implement CellposeModel.eval option"):
        model.segment(index_img, augment=True)
    classical = SegmentationModel(backend="classical")
    img16 = synth.synth_fov(3, size=256)[1]
    assert classical.segment(img16).dtype == np.int64
    with pytest.raises(RuntimeError, match="does not use flow_threshold"):
        classical.segment(img16, flow_threshold=0.8)
    with pytest.raises(RuntimeError, match="takes no CellposeModel.eval options"):
        classical.segment(img16, min_size=30)


def test_config5_at_its_stated_size():
    """BASELINE configs[4] at its own size: the bf16 stand-in network on 2-channel 1024 x 1024 tiles and the HIP
    post-processing (flow filter included) on flow fields of that size; masks / oracle identity on a 1024 x 1024
    synthetic flow field."""
    import torch

    from arcadia_microscopy_tools_amd import cellpose_hip as ch, hipops, synth
    from arcadia_microscopy_tools_amd.device import get_context
    from oracle import cellpose_dynamics as cd

    ctx = get_context()
    dev = torch.device("cuda", 0)
    net, dt = ch.prepare_network(ch.make_standin(), dev, "bf16")
    x = torch.randn(2, 2, 1024, 1024, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = net(x)
    assert y.shape == (2, 3, 1024, 1024) and y.dtype == torch.bfloat16 and bool(torch.isfinite(y.float()).all())
    labels, counts = ch.flows_to_masks(y.float(), niter=20, max_seeds=262144, flow_threshold=0.4, fill_holes=True)
    lab = labels.numpy()
    cnt = counts.numpy()
    assert lab.shape == (2, 1024, 1024) and (cnt >= 0).all()
    for i in range(2):  # labels are 1..count without gaps, whatever the (meaningless) random-weight flows produced
        present = np.unique(lab[i])
        assert present[0] >= 0 and int(present[-1]) == int(cnt[i]) and len(present) == int(cnt[i]) + (1 if present[0] == 0 else 0)
    # a real flow field of the stated size: ~600 disks, 200 iterations, default thresholds -> identical to the oracle
    dP, prob, truth = synth.synthetic_flows((1024, 1024), 600, seed=21, noise=0.3)
    want = cd.compute_masks(dP, prob, niter=200, flow_threshold=0.4, fill_holes=True)
    lab, cnt = hipops.cellpose_masks(ctx.asarray(dP[None]), ctx.asarray(prob[None]), niter=200, flow_threshold=0.4,
                                     fill_holes=True)
    assert int(cnt.numpy()[0]) == int(want.max()) > 300
    assert np.array_equal(lab.numpy()[0], want)


def test_fused_standin_forward_equals_the_eager_one():
    """cellpose_hip.FusedStandIn (the stand-in's forward with batch norm / ReLU / additions / upsampling fused into
    amt_nn_affine_act_bf16 passes) against the eager bf16 forward and a float32 forward of the same weights: the fused
    pass rounds to bf16 once per fused group instead of once per operation, so it must be at least about as close to
    float32 as the eager bf16 pass is.  Also the fused kernel by itself against its definition."""
    import torch

    from arcadia_microscopy_tools_amd import cellpose_hip as ch
    from arcadia_microscopy_tools_amd.device import Context

    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    ref32 = ch.make_standin().to(dev).float().eval()
    for m in ref32.modules():  # running statistics that are not the identity, so that the folded batch norm is exercised
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.6, 1.4)
            m.weight.data.uniform_(0.7, 1.3)
            m.bias.data.uniform_(-0.2, 0.2)
    import copy

    net, dt = ch.prepare_network(copy.deepcopy(ref32), dev, "bf16")
    fused = ch.FusedStandIn(net, Context(0))
    x32 = torch.randn(2, 2, 256, 256, device=dev)
    x = x32.to(dt).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        want32 = ref32(x.float())
        eager = net(x).float()
    got = fused(x)
    torch.cuda.synchronize()
    assert got.shape == (2, 3, 256, 256) and got.dtype == torch.bfloat16
    got = got.float()
    scale = float(want32.abs().mean())
    err_eager = float((eager - want32).abs().mean()) / scale
    err_fused = float((got - want32).abs().mean()) / scale
    assert err_fused < 0.05 and err_fused <= 1.25 * err_eager + 1e-3, (err_fused, err_eager)
    # the kernel against its definition: upsampled input + skip + style, affine map, ReLU; and the x + y side output
    bn = torch.nn.BatchNorm2d(64).to(dev).eval()
    bn.running_mean.uniform_(-1, 1), bn.running_var.uniform_(0.5, 2), bn.weight.data.uniform_(0.5, 2), bn.bias.data.uniform_(-1, 1)
    xs = torch.randn(2, 64, 8, 12, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    ys = torch.randn(2, 64, 16, 24, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    st = torch.randn(2, 64, device=dev)
    pre = torch.randn(64, device=dev)
    out, ssum = fused._glue((xs, pre), bn, True, yb=(ys, None), style=st, upsample=True, want_sum=True)
    torch.cuda.synchronize()
    up = torch.nn.functional.interpolate(xs.float(), scale_factor=2, mode="nearest")
    s_ref = up + ys.float() + pre[None, :, None, None]
    o_ref = torch.relu(bn(s_ref + st[:, :, None, None]))
    assert torch.equal(ssum.float(), s_ref.to(dt).float())
    assert torch.allclose(out.float(), o_ref.to(dt).float(), rtol=2 ** -7, atol=1e-6)
