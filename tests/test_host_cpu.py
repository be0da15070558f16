"""CPU-only checks: the C ABI exports what include/fc3d.h declares, host-side logic of the
drop-in modules, the U-Net restatement, and the data-parallel helpers under gloo (2 ranks)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from fc3d import _native
    hdr = open(os.path.join(ROOT, "include", "fc3d.h")).read()
    declared = set(re.findall(r"\b(fc3d_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = _native.load()                       # loads without a GPU: no HIP call is made
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in fc3d.h but not exported"
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    assert lib.fc3d_version() == _native.ABI_VERSION == 212
    assert lib.fc3d_status_string(-1).decode().startswith("invalid")
    # argument validation happens before any device work
    assert lib.fc3d_visible_mask(None, None, 1, 4, 4, 0, None, None) == -1
    # one 64-bit key per tile/strip of whichever kernel family has the most of them at this size:
    # 8 x 128 for the 64-column x 4-row wavefronts of the large-PSF kernels
    assert lib.fc3d_code_pattern_fwd_workspace(32, 512, 512) == 32 * 1024 * 8
    # tap index: 8-byte boxes, per 512 target pixels one for the group, 4 for its runs, 32 for their segments
    assert lib.fc3d_bilinear_bwd_img_workspace(2, 64, 96, 1, 64, 96) == 2 * 12 * 37 * 8
    assert lib.fc3d_bilinear_bwd_img_workspace(2, 64, 96, 3, 64, 96) == 0            # one-channel images only
    assert lib.fc3d_bilinear_bwd_img_workspace(1, 40000, 8, 1, 8, 8) == 0            # beyond the 16-bit box fields
    # the entry points around the path: one partial maximum per 4096 pixels (at most 256 per image),
    # one float4 partial per 256 pixels
    assert lib.fc3d_capture_post_workspace(32, 512, 512) == 32 * 64 * 4
    assert lib.fc3d_capture_post_workspace(1, 4096, 4096) == 256 * 4
    assert lib.fc3d_reproj_loss_workspace(2, 64, 64) == 2 * 16 * 16
    assert lib.fc3d_capture_post_workspace(0, 4, 4) == 0 and lib.fc3d_reproj_loss_workspace(1, 0, 4) == 0
    one = 16                                    # any non-null value: arguments are checked before they are used
    assert lib.fc3d_box_filter(None, one, 1, 8, 8, 4, 0, 0.0, one + 64) == -1     # even window
    assert lib.fc3d_box_filter(None, one, 1, 8, 8, 35, 0, 0.0, one + 64) == -1    # window beyond 33
    assert lib.fc3d_box_filter(None, one, 1, 8, 8, 5, 3, 0.0, one + 64) == -1     # unknown mode
    assert lib.fc3d_box_filter(None, one, 1, 8, 8, 5, 0, 0.0, one) == -1          # in place
    assert lib.fc3d_capture_post(None, one, one, one, one, one, 1, 8, 8, one, one, None) == -1
    assert lib.fc3d_capture_warp_fwd(None, one, one, one, one, one, one, one, 1, 8, 8, one, None, one, 0, None, None) == -1
    assert lib.fc3d_reflectance_fwd(None, one, -1, one, 1, 8, 8, one) == -1            # negative batch stride
    assert lib.fc3d_reproj_loss_fwd(None, one, one, None, one, one, one, one, one, 1, 8, 8, 8, 8, one, one, one) == -1
    assert lib.fc3d_reproj_loss_fwd_static(None, one, one, None, one, one, one, one, one, 1, 8, 8, 8, 8, one, one) == -1
    # one image too large for the 32-bit per-image byte offsets of the sampling kernels: refused, not wrapped
    big = (32768, 16384)                        # 2^29 pixels
    assert lib.fc3d_warp_p2c_fwd(None, one, one, 0, 0, 0, 0, one, 0, 0, 0, 0, one, one, 1, *big, one, one + 8, one + 16,
                                 one, 0, None) == -2
    assert lib.fc3d_capture_warp_fwd(None, one, one, one, one, one, one, one, 1, *big, one, one, one, 0, None, None) == -2
    assert lib.fc3d_bilinear_bwd_img(None, one, one + 8, None, 1, 8, 8, 1, *big, None, 0, one) == -2
    # the entry points added for the six-launch step and the five-launch capture: null arguments, sizes that
    # do not fit, PSF shapes without a kernel -- all refused before any device work
    from fc3d._native import LayerGrid
    lg = LayerGrid(1.0, 1.0, 1.0, 0.0, 21)
    nk = __import__("ctypes").c_int(0)
    assert lib.fc3d_warp_p2c_pose_fwd(None, one, one, 0, 0, 0, 0, one, 0, 0, 0, 0, None, one, 0, 1, 8, 8, one, one + 8,
                                      one + 16, one, 0, 0, None, one, one, None) == -1        # no pose
    assert lib.fc3d_warp_p2c_pose_fwd(None, one, one, 0, 0, 0, 0, one, 0, 0, 0, 0, one, one, 0, 1, *big, one, one + 8,
                                      one + 16, one, 0, 0, None, one, one, None) == -2
    assert lib.fc3d_warp_project_splat(None, one, 0, 0, 0, 0, one, None, 0, 1, 8, 8, one + 8, one, 0, one, one, None) == -1
    assert lib.fc3d_warp_project_splat(None, one, 0, 0, 0, 0, one, one, 0, 1, *big, one + 8, one, 0, one, one, None) == -2
    assert lib.fc3d_warp_sample(None, one, one, 0, 0, 0, 0, None, 1, 8, 8, one, 0, one, one + 8, None) == -1
    assert lib.fc3d_warp_sample(None, one, one, 0, 0, 0, 0, one, 1, *big, one, 0, one, one + 8, None) == -2
    assert lib.fc3d_bilinear_bwd_img_dot_count(32, 512, 512) == 8 * 43 and lib.fc3d_bilinear_bwd_img_dot_count(1, 0, 5) == 0
    # small problems are cut into shorter tiles: the tallest of 12 / 6 / 4 rows that gives 1,024 workgroups, else 4
    assert lib.fc3d_bilinear_bwd_img_dot_count(8, 512, 512) == 8 * 86 and lib.fc3d_bilinear_bwd_img_dot_count(1, 256, 256) == 4 * 64
    assert lib.fc3d_bilinear_bwd_img_dot(None, one, one + 8, 1, 8, 8, 8, 8, one + 8, None, one, one) == -1      # no image
    assert lib.fc3d_bilinear_bwd_img_dot(None, one, one + 8, 1, 8, 8, *big, one + 8, one, one, one) == -2
    assert lib.fc3d_code_pattern_bwd_dot_supported(5, 5, 1) == 1 and lib.fc3d_code_pattern_bwd_dot_supported(23, 23, 1) == 1
    assert lib.fc3d_code_pattern_bwd_dot_supported(8, 8, 1) == 0 and lib.fc3d_code_pattern_bwd_dot_supported(5, 3, 1) == 0
    assert lib.fc3d_code_pattern_bwd_dot_supported(1, 1, 0) == 1
    assert lib.fc3d_code_pattern_bwd_dot(None, one, one, 4, one, one, one, one, 1, 8, 8, 21, 8, 8, 1, one) == -1  # even PSF
    assert lib.fc3d_code_pattern_bwd_dot(None, one, None, 4, one, one, one, one, 1, 8, 8, 21, 5, 5, 1, one) == -1
    assert lib.fc3d_code_pattern_shaded_conv_fwd(None, one, 0, 8, 1, one, one, 1, 8, 8, 9, 9, lg, one, None, one,
                                                 __import__("ctypes").byref(nk)) == -1                           # 9x9: no rolling-row kernel
    assert lib.fc3d_capture_fwd_workspace(32, 512, 512) == (32 * 32 + 32 * 512) * 4 and lib.fc3d_capture_fwd_workspace(0, 1, 1) == 0
    assert lib.fc3d_capture_fwd(None, one, one, one, one, one, None, one, 0, one, one, one, 1, 8, 8, one, one, one, one,
                                one, 0, 0, one, None, None) == -1
    assert lib.fc3d_capture_fwd(None, one, one, one, one, one, one, one, 0, one, one, one, 1, *big, one, one, one, one,
                                one, 0, 0, one, None, None) == -2
    # more partial sums than the library's static array holds: refused before any device work
    assert lib.fc3d_reproj_loss_fwd_static(None, one, one, one, one, one, one, one, one, 300, 512, 512, 512, 512,
                                           one, one) == -4


def test_layer_grid_matches_reference_tables(golden):
    import optics_pytorch as O
    g = golden("g1_layers")
    z = g["z"].reshape(-1).astype(np.float32)
    for L in (21, 41, 20, 5):
        lg = O._layer_grid(L)
        # the kernel's fp32 recipe, evaluated with numpy fp32 scalars
        c, iz0, st, c0 = (np.float32(v) for v in (lg.coeff, lg.inv_z0, lg.step, lg.centre0))
        r = (np.float32(1.0) / z).astype(np.float32)
        wm = (c * (r - iz0).astype(np.float32)).astype(np.float32)
        rr = np.rint((wm / st).astype(np.float32))
        lf = np.rint((rr - c0).astype(np.float32))
        ok = (lf >= 0) & (lf < L) & (np.abs(rr - (c0 + lf).astype(np.float32)) < np.float32(0.01))
        lay = np.where(ok, lf, -1).astype(np.int16)
        assert np.array_equal(lay, g[f"layer_L{L}"].reshape(-1)), L


def test_dropin_module_surface(golden):
    """Same public names / defaults as the reference modules (train_pytorch.py:23-29 imports)."""
    import inspect
    import geometry_pytorch as G
    import optics_pytorch as O
    for name in ("warp_p2c", "z2pointcloud", "bilinear_sampler", "gen_visible_mask", "coord0TOgrid1",
                 "pixel_to_ray_array", "points_in_camera_coords", "zp_cView_to_zc", "coord_transform",
                 "get_intrinsic", "pixel_to_ray"):
        assert callable(getattr(G, name)), name
    assert (G.H, G.W) == (800, 1200)
    for name in ("codePattern", "codePattern_simple", "gen_WmMask", "z2Wm"):
        assert callable(getattr(O, name)), name
    assert (O.wvl, O.z_min, O.z_max, O.z0) == (530e-9, 0.695714566, 0.941062289, 0.8)
    assert O.PSFs_np.shape == (21, 5, 5, 1) and O.fn_psfs == 'psfs_exp.npy'
    sig = inspect.signature(O.codePattern)
    assert list(sig.parameters)[:4] == ["pattern", "zMap", "device", "is_conv_psf"]
    assert sig.parameters["device"].default == 'cuda' and sig.parameters["is_conv_psf"].default is True
    assert inspect.signature(G.gen_visible_mask).parameters["method"].default == 'round'
    assert list(inspect.signature(G.pixel_to_ray_array).parameters.values())[0].default == 1200
    g0 = golden("g0_rays")
    assert np.abs(G.pixel_to_ray_array(24, 10) - g0["rays_24x10"]).max() < 1e-15
    assert G.pixel_to_ray_array() is G.pixel_to_ray_array()            # cached, not a 1 s python loop
    assert np.abs(np.array([G.pixel_to_ray((0, 0)), G.pixel_to_ray((1199, 799)), G.pixel_to_ray((600, 400))])
                  - g0["corner_rays"]).max() < 1e-15
    assert np.array_equal(G.get_intrinsic(2, device='cpu').numpy(), g0["intrinsic"])
    assert float(O.z2Wm(0.8)) == 0.0 and abs(float(O.z2Wm(O.z_min)) - 60) < 1e-6
    g1 = golden("g1_layers")
    assert np.array_equal(O.z2Wm(torch.from_numpy(g1["z"])).numpy(), g1["Wm"])
    layers = np.linspace(O.z2Wm(O.z_max), O.z2Wm(O.z_min), 21)
    m = O.gen_WmMask(torch.from_numpy(g1["Wm"]), layers, 'cpu')
    idx = m.argmax(-1)
    idx[m.sum(-1) == 0] = -1
    assert np.array_equal(idx.numpy().astype(np.int16), g1["layer_L21"])
    t = torch.rand(1, 4, 4, 3)
    assert torch.equal(O.codePattern_simple(t, None), t[..., 0:1]) and O.codePattern_simple(t[..., :1], None) is not None
    with pytest.raises(RuntimeError, match="GPU only"):
        O.codePattern(torch.zeros(1, 8, 8, 21), torch.ones(1, 8, 8, 1))
    with pytest.raises(UnboundLocalError):
        G.gen_visible_mask(torch.zeros(1, 2, 2, 2), "nearest")
    # the kernels around the path refuse host tensors just as loudly (no silent tensor-op stand-in)
    from fc3d import ops
    one, img = torch.ones(1, 8, 8, 1), torch.rand(1, 8, 8, 1)
    eye = torch.eye(4).unsqueeze(0)
    for call in (lambda: ops.reflectance(img, one),
                 lambda: ops.box_filter(img, 5, "mean"),
                 lambda: ops.capture_post(img, torch.rand(1), img, torch.rand(1), torch.zeros(1, 8, 8, dtype=torch.uint8)),
                 lambda: ops.capture_warp(img, one, one, torch.rand(8), torch.rand(8), eye, eye),
                 lambda: ops.ReprojLossFn.apply(one, one, torch.rand(1, 8, 8, 2), eye, eye, img, img, one)):
        with pytest.raises(RuntimeError, match="GPU only"):
            call()


def test_producers_and_postprocessing_match_reference_fixture(golden):
    """The tensor-op forms of the pattern generator, shading, local normalisation, erosion and
    the masked losses (fc3d.synth / fc3d.capture) against outputs of the reference's utils_pytorch."""
    from fc3d import capture, synth
    g = golden("g8_utils")
    t = torch.from_numpy
    bits = lambda a: np.packbits(a.astype(np.uint8).reshape(-1))  # noqa: E731
    assert np.array_equal(bits(synth.dot_pattern(200, 300, 100, 150)), g["pat_small_bits"])
    assert np.array_equal(bits(synth.dot_pattern(800, 1200)), g["pat_native_bits"])       # default grid = 100 x 150 there
    assert bool(g["pat_small_ch1_equal"])                                                 # every layer holds the same image
    dy, dx = capture.image_gradients(t(g["z"]))
    assert torch.equal(dy, t(g["dy"])) and torch.equal(dx, t(g["dx"]))
    assert torch.equal(capture.with_reflectance(t(g["I"]), t(g["z"])), t(g["refl"]))
    assert torch.equal(capture.local_norm(t(g["Im"]), 17, 0.01), t(g["ln17"]))
    assert torch.equal(capture.local_norm(t(g["Im"]), 5, 0.02), t(g["ln5"]))
    assert torch.equal(capture.erode(t(g["mask"]), 5), t(g["erode5"]))
    for name, fn in (("rms", capture.cost_rms_mask), ("l1", capture.cost_L1_mask), ("grad", capture.cost_grad_mask)):
        hat = t(g["hat"]).clone().requires_grad_(True)
        v = fn(t(g["GT"]), hat, t(g["mask"]))
        v.backward()
        assert abs(float(v) - float(g["loss_" + name])) <= 1e-6 * abs(float(g["loss_" + name])), name
        assert float((hat.grad - t(g["grad_" + name])).abs().max()) <= 1e-6 * float(np.abs(g["grad_" + name]).max()), name


def test_unets_match_reference_parameter_count():
    from fc3d import nets
    nxy = sum(p.numel() for p in nets.UNet_xy().parameters())
    nz = sum(p.numel() for p in nets.UNet_z().parameters())
    assert (nxy, nz) == (8635846, 8635811)                 # 17,271,657 in total, as the reference
    net = nets.UNet_z().eval()
    out = net(torch.rand(1, 32, 48, 1))
    assert out.shape == (1, 32, 48, 1) and float(out.min()) >= nets.Z_MIN and float(out.max()) <= nets.Z_MAX
    out = nets.UNet_xy().eval()(torch.rand(1, 32, 48, 1))
    assert out.shape == (1, 32, 48, 2)


@pytest.mark.parametrize("tag", ["xy", "z"])
def test_unets_load_reference_checkpoints(tag, tmp_path, golden):
    """State-dict keys, shapes, seeded initial weights and outputs of the U-Nets against the reference's
    (network_pytorch.py:56-211, recorded in g10_nets.npz), and a checkpoint round trip through the
    weights-only loader in the layout train_pytorch.py:401-407 writes and :257-259 reads."""
    from fc3d import nets
    g = golden("g10_nets")
    torch.manual_seed(77)
    net = {"xy": nets.UNet_xy, "z": nets.UNet_z}[tag]()
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[tag + "_keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(v) for v in g[tag + "_shapes"]]
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    assert np.allclose(sums, g[tag + "_sums"], rtol=1e-12, atol=1e-12)      # same seed, same order: same weights
    x = torch.from_numpy(g["x"])
    net.train()
    out = net(x)
    assert float((out - torch.from_numpy(g[tag + "_out_train"])).abs().max()) <= 1e-6
    net.eval()
    with torch.no_grad():
        out = net(x)
    assert float((out - torch.from_numpy(g[tag + "_out_eval"])).abs().max()) <= 1e-6
    path = str(tmp_path / "ckpt.pth")
    torch.save({"epoch": 3, "model_" + tag: net.state_dict()}, path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    torch.manual_seed(1)
    other = {"xy": nets.UNet_xy, "z": nets.UNet_z}[tag]()
    other.load_state_dict(ck["model_" + tag], strict=True)
    other.eval()
    with torch.no_grad():
        assert torch.equal(other(x), out)


def test_synthetic_scene_is_deterministic_and_in_range():
    from fc3d import synth
    a = synth.make_scene(2, 32, 48, 5)
    b = synth.make_scene(2, 32, 48, 5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["z_p"].min() >= synth.Z_MIN and a["z_p"].max() <= synth.Z_MAX
    R = a["pose_p2c"][0, :3, :3].astype(np.float64)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-6
    p = synth.make_psfs(21, 5, "disc")
    assert p.shape == (21, 5, 5, 1) and np.allclose(p.sum(axis=(1, 2, 3)), 1, atol=1e-6)
    d = synth.dot_pattern(64, 96)
    assert set(np.unique(d)) == {0.0, 1.0} and 0.05 < d.mean() < 0.5


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(sys.argv[1], "depth-estimation_amd"))
from fc3d import dist as fdist
rank, world, local = fdist.init_from_env("gloo")
assert world == 2
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
bucket = fdist.FlatGradBucket(net.parameters())
assert bucket.nbytes == 4 * sum(p.numel() for p in net.parameters())
x = torch.arange(12.0).reshape(2, 6) * (rank + 1) / 10
bucket.zero()
net(x).sum().backward()
local_grad = bucket.flat.clone()
bucket.average(async_op=True); bucket.wait()
# reference: gather every rank's local gradient and average by hand
allg = [torch.zeros_like(local_grad) for _ in range(world)]
torch.distributed.all_gather(allg, local_grad)
assert torch.allclose(bucket.flat, sum(allg) / world, atol=1e-7)
assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())   # still views of the bucket
assert fdist.max_over_ranks(float(rank + 1)) == 2.0
assert fdist.shard_seed(1000, rank) == 1000 + 100 * rank

# --- replicas start from rank 0's weights whatever their own seed was
torch.manual_seed(100 + rank)
conv = lambda: torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(4, 4, 3, padding=1),
                                   torch.nn.Tanh(), torch.nn.Conv2d(4, 1, 3, padding=1))
net = conv()
fdist.broadcast_module_state([net])
w0 = [p.detach().clone() for p in net.parameters()]
g0 = [torch.zeros_like(w) for w in w0]
for w, g in zip(w0, g0):
    g.copy_(w); torch.distributed.broadcast(g, 0); assert torch.equal(g, w)

# --- three ranges, all-reduces started from autograd hooks while backward is still running:
#     the averaged gradient of per-rank mean losses == the gradient of the mean loss over the
#     concatenated batch in one process
gen = torch.Generator().manual_seed(7)
X = torch.rand(4, 1, 8, 8, generator=gen); Y = torch.rand(4, 1, 8, 8, generator=gen); M = (torch.rand(4, 1, 8, 8, generator=gen) > 0.3).float()
single = conv(); single.load_state_dict(net.state_dict())
(single(X) - Y).square().mean().backward()
bucket = fdist.FlatGradBucket(net.parameters(), n_buckets=3, overlap=True)
assert len(bucket.ranges) == 3 and bucket.ranges[0][0] == 0 and bucket.ranges[-1][1] == bucket.flat.numel()
sl = slice(2 * rank, 2 * rank + 2)
for _ in range(2):                     # twice: the per-range counters re-arm
    bucket.zero()
    (net(X[sl]) - Y[sl]).square().mean().backward()
    assert len(bucket._handles) == 3   # every range went out during backward
    bucket.average(async_op=True); bucket.wait()
    for p, q in zip(net.parameters(), single.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-7), (p.grad - q.grad).abs().max()

# --- the reference's masked losses are global sums over global counts (utils_pytorch.py:161-163):
#     with the sums added over ranks every rank sees the single-process loss, and the SUM of the
#     ranks' gradients is its gradient
def masked_rms(gt, hat, m, reduce):
    return torch.sqrt(reduce(torch.sum(torch.square(m * (gt - hat)))) / reduce(torch.sum(m)).clamp(min=1e-6) + 1e-6)
single.zero_grad()
Ls = masked_rms(Y, single(X), M, lambda v: v); Ls.backward()
bucket.close()                         # the hooks of the three-range bucket go; the next one registers its own
bucket = fdist.FlatGradBucket(net.parameters(), n_buckets=2, overlap=True, average=False)
bucket.zero()
Lg = masked_rms(Y[sl], net(X[sl]), M[sl], fdist.sum_over_ranks); Lg.backward()
bucket.average(); 
assert abs(float(Lg) - float(Ls)) < 1e-6
for p, q in zip(net.parameters(), single.parameters()):
    assert torch.allclose(p.grad, q.grad, atol=1e-6), (p.grad - q.grad).abs().max()
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("rank", rank, "ok")
'''


def test_data_parallel_helpers_gloo_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script), ROOT],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


def test_bucket_refuses_a_second_backward_without_zero():
    """With overlap=True a range's all-reduce starts from the hook that completes it; a second backward
    without zero() would add onto values that are already being summed over ranks.  The hook raises instead."""
    from fc3d import dist as fdist
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Tanh(), torch.nn.Linear(3, 1))
    x = torch.rand(5, 4)
    bucket = fdist.FlatGradBucket(net.parameters(), n_buckets=2, overlap=True)
    bucket.zero()
    net(x).sum().backward()
    g1 = bucket.flat.clone()
    with pytest.raises(RuntimeError, match="zero\\(\\) before every backward"):
        net(x).sum().backward()
    bucket.zero()
    net(x).sum().backward()
    assert torch.equal(bucket.flat, g1)
    bucket.close()
    # without overlap there are no hooks: accumulating over two backward passes is allowed
    acc = fdist.FlatGradBucket(net.parameters(), n_buckets=2, overlap=False)
    acc.zero()
    net(x).sum().backward()
    net(x).sum().backward()
    assert torch.allclose(acc.flat, 2 * g1)


def test_bench_starts_its_own_ranks_when_invoked_plainly():
    """`python bench.py --gpus 2` without torch.distributed.run's environment starts two ranks as child
    processes and relays rank 0's line.  --launch-selftest runs the rendezvous / barrier / max-over-ranks /
    reporting path with gloo and a host stand-in for the step: no GPU call is made anywhere."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest", "--steps", "8",
                        "--warmup", "2", "--n1-value", "1000"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly one line, from rank 0
    d = json.loads(lines[0])
    assert d["selftest"] is True and d["n_gpus"] == 2 and d["world_size"] == 2 and d["steps"] == 8
    assert d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    pr = d["per_rank_captures_per_s"]
    assert 0 < pr["min"] <= pr["max"]
    assert d["value"] <= 2 * pr["max"] * 1.0001            # whole-job rate: never above the sum of the ranks' own rates
    assert abs(d["scaling_efficiency_vs_n1"] - d["value"] / 2000.0) < 1e-9
    # a mismatch between --gpus and the environment of an outer launcher is refused, not silently run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
    from fc3d import launch
    cmd = launch.rank_command("x.py", ["--gpus", "4"], 4, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-2:] == ["--gpus", "4"]


def test_capture_data_shards(tmp_path):
    """The .npz reader: stored values passed through as float32 like the reference's loader
    (data_provider_pytorch.py:32-36), optional decoding of raw uint16 codes (value/50000 + 0.3), split
    discovery, per-rank shares that are disjoint, equally long and cover whole batches only, seeded
    shuffling, memory-mapped and compressed archives."""
    from fc3d import data, synth
    sc = synth.make_scene(11, 16, 24, 5)
    root = str(tmp_path) + os.sep
    data.write_npz(root + "train_a.npz", sc["z_p"], sc["z_c"], sc["pose_p2c"], as_uint16=True)
    data.write_npz(root + "test_a.npz", sc["z_p"], sc["z_c"], sc["pose_p2c"])
    with pytest.raises(FileNotFoundError):
        data.read_data(root, 2, "valid")
    codes = np.load(root + "train_a.npz")["z_p"]
    assert codes.dtype == np.uint16
    assert np.array_equal(data.decode_depth(codes, True), codes.astype(np.float32) / np.float32(50000) + np.float32(0.3))
    assert np.abs(data.decode_depth(codes, True) - sc["z_p"]).max() <= 0.5 / 50000 + 1e-7
    # default: the stored values themselves, as float32 -- what the reference's __getitem__ returns
    raw = next(iter(data.read_data(root, 2, "train", seed=3)))
    assert raw["z_p"].dtype == torch.float32 and float(raw["z_p"].min()) > 1000
    assert np.array_equal(data.decode_depth(codes), codes.astype(np.float32))
    assert isinstance(data.open_member(root + "train_a.npz", "z_p"), np.memmap)         # np.savez stores uncompressed
    np.savez_compressed(root + "test_c.npz", z_p=sc["z_p"], z_c=sc["z_c"], pose_p2c=sc["pose_p2c"])
    assert not isinstance(data.open_member(root + "test_c.npz", "z_p"), np.memmap)
    assert np.array_equal(data.open_member(root + "test_c.npz", "z_c"), sc["z_c"])
    seen = []
    for rank in range(2):
        sh = data.read_data(root, 2, "train", rank=rank, world=2, seed=3, decode_uint16=True)
        assert float(next(iter(data.read_data(root, 2, "train", rank=rank, world=2, decode_uint16=True)))["z_p"].max()) < 1.0
        assert len(sh) == 2 and len(sh.index) == 4            # 11 samples -> 5 per rank -> 2 whole batches of 2
        batches = list(sh)
        assert len(batches) == 2 and tuple(batches[0]["z_p"].shape) == (2, 16, 24, 1)
        assert batches[0]["z_p"].dtype == torch.float32 and tuple(batches[0]["pose_p2c"].shape) == (2, 4, 4)
        again = [b["pose_p2c"] for b in data.read_data(root, 2, "train", rank=rank, world=2, seed=3, decode_uint16=True)]
        assert all(torch.equal(a, b["pose_p2c"]) for a, b in zip(again, batches))   # same seed, same order
        seen.append(set(sh.index.tolist()))
    assert not (seen[0] & seen[1]) and all(i % 2 == r for r in range(2) for i in seen[r])
    seq = data.read_data(root, 3, "test")
    first = next(iter(seq))
    assert np.array_equal(first["z_p"].numpy(), sc["z_p"][:3]) and np.array_equal(first["pose_p2c"].numpy(), sc["pose_p2c"][:3])


def test_library_is_rebuilt_when_sources_change_not_when_times_do(tmp_path, monkeypatch):
    """The build check compares a hash of the sources, the header and the flags with the one stored beside
    the library when it was built: file times play no part (a copied or checked-out tree has arbitrary ones)."""
    from fc3d import build
    if not os.path.exists(build.STAMP):
        pytest.skip("library was not built through fc3d.build in this tree")
    assert not build.needs_build()
    os.utime(os.path.join(build.CSRC, "optics.hip"))              # newer than the library now: still up to date
    assert not build.needs_build()
    h = build.source_hash()
    assert h == open(build.STAMP).read().strip() and h != build.source_hash(("-DFC3D_NT=0",))
    monkeypatch.setattr(build, "STAMP", str(tmp_path / "other.stamp"))
    assert build.needs_build()                                     # no record of what the library was built from
    (tmp_path / "other.stamp").write_text("0" * 64 + "\n")
    assert build.needs_build()                                     # built from something else


def test_missing_library_fails_loudly(monkeypatch):
    """No fallback: without the shared library the binding refuses to load."""
    from fc3d import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", os.path.join(ROOT, "does", "not", "exist", "libfc3d_hip.so"))
    with pytest.raises(ImportError, match="no fallback"):
        _native.load()


def test_points_note_is_dropped_when_it_no_longer_holds():
    """The note that lets warp_p2c read a depth map instead of the points made from it (host logic only, CPU
    tensors): honoured for the plain permuted view of untouched points, dropped for copies, other views,
    in-place writes to either tensor, points that require a gradient, and when switched off."""
    import torch
    from fc3d import ops
    B, H, W = 2, 5, 7
    z = torch.rand(B, H, W, 1)
    rx, ry = torch.rand(W), torch.rand(H)
    pts = torch.rand(B, H, W, 4)
    ops.note_points_from_depth(pts, z, rx, ry)
    view = pts.permute(0, 3, 1, 2)
    got = ops._points_note(view)
    assert got is not None and got[0] is z and got[1] is rx and got[2] is ry
    assert ops._points_note(view.clone()) is None and ops._points_note(view.contiguous()) is None
    assert ops._points_note(view[:, :, 1:]) is None and ops._points_note(pts.permute(0, 3, 2, 1)) is None
    old = ops.set_points_from_depth(False)
    try:
        assert ops._points_note(view) is None
    finally:
        ops.set_points_from_depth(old)
    assert ops._points_note(view) is not None
    z.add_(1.0)                                   # the depth map changed after the points were made
    assert ops._points_note(view) is None
    pts2 = torch.rand(B, H, W, 4)
    ops.note_points_from_depth(pts2, z, rx, ry)
    v2 = pts2.permute(0, 3, 1, 2)
    assert ops._points_note(v2) is not None
    pts2[..., 0] *= 2.0                           # the points were edited
    assert ops._points_note(v2) is None
    pts3 = torch.rand(B, H, W, 4, requires_grad=True)
    ops.note_points_from_depth(pts3, z, rx, ry)
    assert ops._points_note(pts3.permute(0, 3, 1, 2)) is None
