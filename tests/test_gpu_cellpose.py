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
