"""The callers of the path (capture simulator, re-projection loss, one training step) on the GPU,
against the same composition written with the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import geometry_pytorch as G
    import optics_pytorch as O
    from fc3d import capture, synth
    import fc3d_oracle as orc
    return dict(G=G, O=O, capture=capture, synth=synth, orc=orc)


def K_fn(W, H):
    def get_intrinsic(B, device="cuda"):
        K = np.diag([1666.67, 1666.67, 1, 1])
        K[0][2], K[1][2] = W / 2, H / 2
        return torch.from_numpy(K).float().to(device).unsqueeze(0).repeat(B, 1, 1)
    return get_intrinsic


@pytest.mark.parametrize("fused", [True, False], ids=["capture-kernels", "composed"])
def test_forward_model_and_reproj_vs_oracle(env, monkeypatch, fused):
    G, O, capture, synth, orc = (env[k] for k in ("G", "O", "capture", "synth", "orc"))
    B, H, W, L, N = 2, 96, 128, 21, 64
    monkeypatch.setattr(G, "get_intrinsic", K_fn(W, H))
    sc = synth.make_scene(B, H, W, 91)
    psfs = synth.make_psfs(L, 5, "disc")
    O.PSFs_np = psfs
    rs = np.random.RandomState(92)
    noise = dict(offset=rs.rand(B, 1, 1, 1).astype(np.float32), gauss=rs.randn(B, H, W, 1).astype(np.float32),
                 scale=rs.rand(B, 1, 1, 1).astype(np.float32))
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    t = torch.from_numpy
    # dots on a bright background: the capture clears the 0.05 threshold over whole regions, so the
    # eroded mask of the re-projection loss is not empty
    pat_np = (0.5 + 0.5 * synth.dot_pattern(H, W)).astype(np.float32).reshape(1, H, W, 1)
    out = capture.forward_model(d(sc["z_p"]), d(sc["z_c"]), d(sc["pose_p2c"]), L, True, N,
                                {k: d(v) for k, v in noise.items()}, pattern=d(pat_np), fused=fused)
    Ic_s, Ic_m, xyz, xy, zc, zp, pose, zp_full, Ic, Ip_coded, Ic_s_full = out
    assert Ic_s.shape == (B, N, N, 1) and xyz.shape == (B, N, N, 3) and Ip_coded.shape == (B, H, W, 1)

    # the same composition with the oracle on the CPU
    z_p, z_c, pose_c = t(sc["z_p"]), t(sc["z_c"]), t(sc["pose_p2c"])
    Ip = t(pat_np)
    dy = F.pad(z_p[:, 1:] - z_p[:, :-1], (0, 0, 0, 0, 0, 1))
    dx = F.pad(z_p[:, :, 1:] - z_p[:, :, :-1], (0, 0, 0, 1, 0, 0))
    refl = torch.cos(torch.atan(torch.sqrt(dy ** 2 + dx ** 2) / 100e-6)) * 0.8 + 0.2
    pat = (Ip * refl).expand(-1, -1, -1, L)
    Ipc, Ic_r, gp, gc = orc.image_formation(pat, z_p, orc.z2pointcloud(z_p), orc.z2pointcloud(z_c), pose_c, psfs,
                                            W / 2, H / 2)
    Ie = Ic_r + t(noise["offset"]) * 0.05 + t(noise["gauss"]) * 0.005
    Is = torch.clamp(Ie / Ie.amax(dim=(1, 2), keepdim=True) * (t(noise["scale"]) * 0.3 + 0.7), 0, 1)
    vis = orc.gen_visible_mask(gp, "all")
    xy_r = t(orc.pixel_to_ray_array(W, H)[:, :, :2]).float().unsqueeze(0).expand(B, -1, -1, -1)
    xyz_r = orc.bilinear_sampler(torch.cat([xy_r, z_p], -1), gc) * vis
    h0, w0 = (H - N) // 2, (W - N) // 2
    crop = lambda a: a[:, h0:h0 + N, w0:w0 + N]  # noqa: E731
    assert rel_err(Ip_coded.cpu().numpy(), Ipc.numpy()) < 1e-4
    assert rel_err(Ic.cpu().numpy(), crop(Ic_r).numpy()) < 1e-4
    assert rel_err(Ic_s.cpu().numpy(), crop(Is).numpy()) < 1e-4
    assert rel_err(xyz.cpu().numpy(), crop(xyz_r).numpy()) < 1e-4
    m_ref = crop((Is > 0.05).float() * vis).numpy()
    assert (Ic_m.cpu().numpy() != m_ref).mean() < 1e-3        # threshold at 0.05 may flip on a rounding-level tie

    # re-projection loss and its gradient w.r.t. the predicted depth
    torch.manual_seed(94)
    zhat = (zc + 0.003 * torch.randn_like(zc)).requires_grad_(True)
    loss, l_zc, l_rms, l_l1 = capture.reProj_IpGT(zhat, pose, xy, Ic, zc, Ic_m, Ip_coded)
    loss.backward()
    zh = zhat.detach().cpu().requires_grad_(True)
    me = (F.avg_pool2d(Ic_m.cpu().permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1) > 0.999).float()
    zc_r, grid_r, Ic_rp = orc.reproject_chain(zh * me, pose_c, xy.cpu(), Ip_coded.cpu(), W / 2, H / 2)
    l1 = torch.sum(me * torch.abs(Ic.cpu() - Ic_rp)) / (me.sum() + 1)
    lz = torch.sqrt(torch.sum(torch.square(me * (zc.cpu() - zc_r))) / me.sum().clamp(min=1e-6) + 1e-6)
    (l1 + lz).backward()
    assert abs(float(loss) - float(l1 + lz)) < 1e-4 * abs(float(l1 + lz))
    # the L1 term's gradient is +-const per pixel: where |Ic - Ic'| is at rounding level the sign may differ
    # between the two evaluations, so single pixels are allowed to disagree
    gd = (zhat.grad.cpu() - zh.grad).abs()
    assert float((gd > 2e-3 * float(zh.grad.abs().max())).float().mean()) < 1e-3
    assert float(gd.median()) <= 1e-5 * float(zh.grad.abs().max())
    assert float(me.sum()) > 0 and float(zh.grad.abs().max()) > 0


@pytest.mark.parametrize("B,H,W", [(2, 96, 128), (3, 37, 53), (1, 4, 5), (1, 1, 1), (1, 8, 4100), (1, 800, 1200)])
def test_capture_kernels_match_composed_operators(env, monkeypatch, B, H, W):
    """The capture kernels against the operators they stand for, on the same device inputs:
    warp straight from the depth maps == points + warp_p2c + bilinear_sampler (bit for bit, masks
    and grids included); sensor model == the tensor expression evaluated by torch on the CPU
    (IEEE division), bit for bit; shading within 1e-5 (it skips the atan/cos roundings)."""
    G, capture, synth = env["G"], env["capture"], env["synth"]
    from fc3d import ops
    monkeypatch.setattr(G, "get_intrinsic", K_fn(W, H))
    sc = synth.make_scene(B, H, W, 95)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    z_p, z_c, pose = d(sc["z_p"]), d(sc["z_c"]), d(sc["pose_p2c"])
    rs = np.random.RandomState(96)
    Ip = d(rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32))

    # shading
    for pat in (Ip, Ip[:1]):
        got = ops.reflectance(pat, z_p)
        want = capture.with_reflectance(pat, z_p)
        assert got.shape == want.shape
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 1e-5

        # ... and fused with the no-blur codePattern (the training mode): identical to the two steps
        one = env["O"].codePattern(got, z_p, is_conv_psf=False, n_layers=21)
        assert torch.equal(ops.code_pattern_shaded(pat, z_p, env["O"]._layer_grid(21)), one)

    # warp from the depth maps
    rays = G.pixel_to_ray_array(W, H)
    coord_p = G.points_in_camera_coords(z_p, rays).permute(0, 3, 1, 2)
    coord_c = G.points_in_camera_coords(z_c, rays).permute(0, 3, 1, 2)
    Ic_ref, gp_ref, gc_ref = G.warp_p2c(Ip, coord_p, coord_c, pose)
    vis_ref = G.gen_visible_mask(gp_ref, "all")
    xy = capture._rays_xy(H, W, z_p.device).unsqueeze(0).expand(B, -1, -1, -1)
    xyz_ref = G.bilinear_sampler(torch.cat([xy, z_p], dim=-1), gc_ref) * vis_ref
    rx, ry = G._separable_rays(W, H, z_p.device)
    T_p2c, T_c2p, _ = ops.pose_transforms(pose, G._intrinsics_for(B, z_p.device), want_fwd=True, want_inv=True)
    Ic, xyz, mask, gp, gc = ops.capture_warp(Ip, z_p, z_c, rx, ry, T_p2c, T_c2p, want_grids=True)
    assert torch.equal(gp, gp_ref) and torch.equal(gc, gc_ref)
    assert torch.equal(mask.view(B, H, W, 1).float(), vis_ref)
    assert torch.equal(Ic, Ic_ref)
    assert torch.equal(xyz, xyz_ref)
    Ic2, xyz2, mask2 = ops.capture_warp(Ip, z_p, z_c, rx, ry, T_p2c, T_c2p)      # grids not asked for
    assert torch.equal(Ic2, Ic) and torch.equal(xyz2, xyz) and torch.equal(mask2, mask)

    # sensor model
    off, sca = rs.rand(B, 1, 1, 1).astype(np.float32), rs.rand(B, 1, 1, 1).astype(np.float32)
    gau = rs.randn(B, H, W, 1).astype(np.float32)
    scaled, valid = ops.capture_post(Ic, d(off), d(gau), d(sca), mask)
    t = torch.from_numpy
    Ie = Ic.cpu() + t(off) * 0.05 + t(gau) * 0.005
    Is = torch.clamp(Ie / Ie.amax(dim=(1, 2), keepdim=True) * (t(sca) * 0.3 + 0.7), 0.0, 1.0)
    Im = (Is > 0.05).float() * vis_ref.cpu()
    assert torch.equal(scaled.cpu(), Is)
    assert torch.equal(valid.cpu(), Im)


def test_capture_kernels_vs_reference_fixture(env, golden):
    """Shading, local normalisation and erosion kernels against outputs of the reference's
    utils_pytorch (withReflectance, ln, the avg-pool erosion)."""
    from fc3d import ops
    g = golden("g8_utils")
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    assert rel_err(ops.reflectance(d(g["I"]), d(g["z"])).cpu().numpy(), g["refl"]) < 1e-5
    assert rel_err(ops.box_filter(d(g["Im"]), 17, "local_norm", 0.01).cpu().numpy(), g["ln17"]) < 1e-5
    assert rel_err(ops.box_filter(d(g["Im"]), 5, "local_norm", 0.02).cpu().numpy(), g["ln5"]) < 1e-5
    assert np.array_equal(ops.box_filter(d(g["mask"]), 5, "erode").cpu().numpy(), g["erode5"])


@pytest.mark.parametrize("B,H,W", [(2, 37, 53), (1, 64, 128), (3, 5, 3), (1, 100, 70)])
def test_box_filter_vs_avg_pool(env, B, H, W):
    """Local normalisation and mask erosion against F.avg_pool2d on the CPU: erosion of a 0/1
    mask bit for bit, means within rounding."""
    from fc3d import ops
    capture = env["capture"]
    rs = np.random.RandomState(B * 1000 + H)
    img = torch.from_numpy(rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32))
    mask = torch.from_numpy((rs.uniform(0, 1, (B, H, W, 1)) > 0.08).astype(np.float32))
    pool = lambda t, k: F.avg_pool2d(t.permute(0, 3, 1, 2), k, stride=1, padding=k // 2).permute(0, 2, 3, 1)  # noqa: E731
    for k in (1, 3, 5, 9, 17, 33):
        mean = ops.box_filter(img.cuda(), k, "mean").cpu()
        assert rel_err(mean.numpy(), pool(img, k).numpy()) < 1e-5
        ln = capture.local_norm(img.cuda(), k, 0.01).cpu()
        assert rel_err(ln.numpy(), (img / (pool(img, k) + 0.01)).numpy()) < 1e-5
        er = capture.erode(mask.cuda(), k).cpu()
        assert torch.equal(er, (pool(mask, k) > 0.999).float())
    # a cropped (strided) view, as the training step passes
    v = img.cuda()[:, 1:H - 1, 1:W - 1, :]
    if v.numel():
        assert rel_err(capture.local_norm(v, 5).cpu().numpy(), (v.cpu() / (pool(v.cpu(), 5) + 0.01)).numpy()) < 1e-5
    with pytest.raises(RuntimeError):
        ops.box_filter(img.cuda(), 4, "mean")


@pytest.mark.parametrize("B,H,W,N", [(2, 96, 128, 64), (3, 41, 57, None)])
def test_reproj_loss_kernels_match_composed(env, monkeypatch, B, H, W, N):
    """The loss kernels (masked re-projection sums, forward and backward) against the same loss
    composed from the chain operators and tensor reductions: the four values and d loss / d depth."""
    G, O, capture, synth = env["G"], env["O"], env["capture"], env["synth"]
    monkeypatch.setattr(G, "get_intrinsic", K_fn(W, H))
    O.PSFs_np = synth.make_psfs(21, 5, "disc")
    sc = synth.make_scene(B, H, W, 97)
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    # a dense pattern: with the dot array the 5x5 erosion of (Ic > 0.05) leaves nothing to compare
    pat = d(np.random.RandomState(98).uniform(0.5, 1.0, (1, H, W, 1)).astype(np.float32))
    with torch.no_grad():
        out = capture.forward_model(d(sc["z_p"]), d(sc["z_c"]), d(sc["pose_p2c"]), 21, True, N, pattern=pat)
    Ic_s, Ic_m, xyz, xy, zc, zp, pose, _, Ic, Ip_coded, _ = out
    assert float(capture.erode(Ic_m, 5).mean()) > 0.3
    torch.manual_seed(5)
    z0 = xyz[..., 2:3] + 0.003 * torch.randn_like(zc)          # a prediction of the projector-frame depth
    res = {}
    for fused in (True, False):
        zh = z0.clone().requires_grad_(True)
        vals = capture.reProj_IpGT(zh, pose, xy, Ic, zc, Ic_m, Ip_coded, fused=fused)
        (vals[0] + 0.5 * vals[2]).backward()                   # the unused RMS term gets a gradient too
        res[fused] = ([float(v) for v in vals], zh.grad.clone())
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 2e-5 * abs(b), (res[True][0], res[False][0])
    ga, gb = res[True][1], res[False][1]
    assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max())
    assert float(Ic_m.sum()) > 0 and float(gb.abs().max()) > 0


def test_train_step_runs(env, monkeypatch):
    G, O, capture, synth = env["G"], env["O"], env["capture"], env["synth"]
    B, H, W = 2, 64, 64
    monkeypatch.setattr(G, "get_intrinsic", K_fn(W, H))
    O.PSFs_np = synth.make_psfs(21, 5, "disc")
    sc = synth.make_scene(B, H, W, 93)
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    torch.manual_seed(0)
    step = capture.TrainStep(torch.device("cuda"))
    assert step.bucket.nbytes == 17271657 * 4
    w0 = step.params[0].detach().clone()
    losses = [float(step(d(sc["z_p"]), d(sc["z_c"]), d(sc["pose_p2c"]))) for _ in range(3)]
    assert all(np.isfinite(losses)), losses
    assert not torch.equal(w0, step.params[0].detach())               # the optimizer moved the weights
    assert float(step.bucket.flat.abs().sum()) > 0


def test_graph_replay_matches_eager(env, monkeypatch):
    """The HIP-graph replay of the step gives the eager results, follows in-place input updates
    and keeps doing so over many replays."""
    G, O, synth = env["G"], env["O"], env["synth"]
    from fc3d.graphs import GraphedImageFormation
    wl = synth.make_workload("C1", B=2, pattern_kind="uniform")
    cfg = wl["cfg"]
    H, W, L = cfg["H"], cfg["W"], cfg["L"]
    K = np.diag([1666.67, 1666.67, 1, 1])
    K[0][2], K[1][2] = W / 2, H / 2
    Kt = torch.from_numpy(K).float().cuda().unsqueeze(0).repeat(2, 1, 1)
    monkeypatch.setattr(G, "get_intrinsic", lambda B, device="cuda": Kt)
    O.PSFs_np = wl["psfs"]
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    rays = G.pixel_to_ray_array(W, H)
    z_p, z_c, pose, g = d(wl["z_p"]), d(wl["z_c"]), d(wl["pose_p2c"]), d(wl["upstream_grad"])
    coord_p = G.points_in_camera_coords(z_p, rays).permute(0, 3, 1, 2)
    coord_c = G.points_in_camera_coords(z_c, rays).permute(0, 3, 1, 2)

    def eager(pat_np):
        p = d(pat_np).requires_grad_(True)
        Ic, _, _ = G.warp_p2c(O.codePattern(p, z_p, n_layers=L), coord_p, coord_c, pose)
        Ic.backward(g)
        return Ic.detach().clone(), p.grad.clone()

    pat = d(wl["pattern"]).requires_grad_(True)
    gf = GraphedImageFormation(pat, z_p, coord_p, coord_c, pose, g, L)
    Ic_e, gr_e = eager(wl["pattern"])
    for _ in range(3):
        Ic_g, gr_g = gf.replay()
    torch.cuda.synchronize()
    assert torch.equal(Ic_g, Ic_e)
    assert float((gr_g - gr_e).abs().max()) <= 1e-5 * float(gr_e.abs().max())
    new = np.random.RandomState(7).uniform(0, 1, wl["pattern"].shape).astype(np.float32)
    # replays issued right behind other work on the stream, with the inputs changing every time
    for src in (new, new, wl["pattern"], new, wl["pattern"]):
        with torch.no_grad():
            pat.copy_(d(src))
        Ic_g, gr_g = gf.replay()
        Ic_e, gr_e = eager(src)
        torch.cuda.synchronize()
        assert torch.equal(Ic_g, Ic_e)
        assert float((gr_g - gr_e).abs().max()) <= 1e-5 * float(gr_e.abs().max())


def test_graphed_capture_matches_eager(env, monkeypatch):
    """Replaying the capture simulator as one graph gives the eager results, also after the input
    buffers were refilled and with replays issued back to back."""
    G, O, capture, synth = env["G"], env["O"], env["capture"], env["synth"]
    from fc3d.graphs import GraphedCapture
    B, H, W, N = 2, 96, 128, 64
    Kt = K_fn(W, H)(B)                     # made once: an upload inside the captured region cannot be replayed
    monkeypatch.setattr(G, "get_intrinsic", lambda n, device="cuda": Kt)
    O.PSFs_np = synth.make_psfs(21, 5, "disc")
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    scenes = [synth.make_scene(B, H, W, s) for s in (101, 102)]
    rs = np.random.RandomState(103)
    noises = [dict(offset=d(rs.rand(B, 1, 1, 1).astype(np.float32)), gauss=d(rs.randn(B, H, W, 1).astype(np.float32)),
                   scale=d(rs.rand(B, 1, 1, 1).astype(np.float32))) for _ in range(2)]
    z_p, z_c, pose = (d(scenes[0][k]).clone() for k in ("z_p", "z_c", "pose_p2c"))
    noise = {k: v.clone() for k, v in noises[0].items()}
    gc = GraphedCapture(z_p, z_c, pose, noise, 21, True, N)
    for i in (0, 1, 1, 0, 1):
        with torch.no_grad():
            z_p.copy_(d(scenes[i]["z_p"])); z_c.copy_(d(scenes[i]["z_c"])); pose.copy_(d(scenes[i]["pose_p2c"]))
            for k in noise:
                noise[k].copy_(noises[i][k])
            got = gc.replay()
            want = capture.forward_model(d(scenes[i]["z_p"]), d(scenes[i]["z_c"]), d(scenes[i]["pose_p2c"]), 21, True, N,
                                         noises[i])
        torch.cuda.synchronize()
        for j in (0, 1, 2, 8, 9, 10):            # Ic_scaled, Ic_mask, xyz_cView, Ic, Ip_coded, Ic_scaled (full)
            assert torch.equal(got[j], want[j]), (i, j)
