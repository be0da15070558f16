"""The CPU oracle against the fixtures captured from the reference (CPU only).

Integer / mask / index results must be bit-exact; floats must agree to far
better than the 1e-4 the GPU path is held to, because both sides are torch-CPU."""
import numpy as np
import pytest
import torch

import fc3d_oracle as orc
from conftest import rel_err
from fc3d import synth

TIGHT = 2e-6


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def full_pattern(p0, L, seed):
    B, H, W, _ = p0.shape
    rest = np.random.RandomState(seed).uniform(0, 1, (B, H, W, L)).astype(np.float32)
    rest[..., 0:1] = p0
    return rest


def test_rays_and_intrinsic(golden):
    g = golden("g0_rays")
    assert np.abs(orc.pixel_to_ray_array(24, 10) - g["rays_24x10"]).max() < 1e-15
    assert np.abs(synth.ray_table(24, 10) - g["rays_24x10"]).max() < 1e-15
    assert np.array_equal(orc.get_intrinsic(2).numpy(), g["intrinsic"])
    full = orc.pixel_to_ray_array()
    assert np.abs(full[[0, 799, 400], [0, 1199, 600]] - g["corner_rays"]).max() < 1e-15


def test_layers(golden):
    g = golden("g1_layers")
    z = t(g["z"])
    assert np.array_equal(orc.z2Wm(z).numpy(), g["Wm"])
    for L in (21, 41, 20, 5):
        assert np.array_equal(orc.layer_index(z, L)[..., 0].numpy().astype(np.int16), g[f"layer_L{L}"]), L
    # documented anchors: z_min -> top layer, z_max -> layer 0, z0 -> middle, 0.75 -> 14, 0.5 -> none
    li = orc.layer_index(t(np.array([orc.Z_MIN, orc.Z_MAX, 0.8, 0.75, 0.5], np.float32).reshape(1, 1, 5, 1)), 21)
    assert li.flatten().tolist() == [20, 0, 10, 14, -1]
    assert float(np.abs(np.array(g["layer_L20"])).max()) == 1 and (g["layer_L20"] == -1).all()   # even L never matches


def _cases(g):
    for name in g["cases"]:
        tag, B, H, W, psf, K, conv, pk = str(name).split("|")
        yield tag, int(B[1:]), int(H[1:]), int(W[1:]), psf, int(K[1:]), bool(int(conv[4:])), pk


def test_code_pattern_forward_and_grad(golden):
    g = golden("g2_code_pattern")
    n = 0
    for tag, B, H, W, psf, K, conv, pk in _cases(g):
        pat = full_pattern(g[f"{tag}_p0"], 21, int(g[f"{tag}_patseed"]))
        psfs = synth.make_psfs(21, K, psf)
        pt = t(pat).requires_grad_(True)
        out, raw, mx = orc.codePattern(pt, t(g[f"{tag}_z"]), psfs, conv, return_parts=True)
        out.backward(t(g[f"{tag}_g"]))
        assert rel_err(out.detach().numpy(), g[f"{tag}_out"]) < TIGHT, tag
        assert rel_err(pt.grad[..., 0:1].numpy(), g[f"{tag}_grad0"]) < 5e-6, tag
        assert float(pt.grad[..., 1:].abs().max()) == 0.0
        # out-of-range depth -> exactly zero output
        li = orc.layer_index(t(g[f"{tag}_z"]), 21)
        assert float(out.detach()[li < 0].abs().max()) == 0.0
        n += 1
    assert n >= 20


def test_argmax_rule_matches_autograd():
    """The max's gradient lands on: smallest w among maxima, then smallest h."""
    raw = torch.zeros(1, 6, 7, 1)
    for (h, w) in ((4, 2), (1, 2), (0, 5), (3, 3)):
        raw[0, h, w, 0] = 1.0
    r = raw.clone().requires_grad_(True)
    m = torch.max(torch.max(r, dim=1, keepdim=True)[0], dim=2, keepdim=True)[0]
    m.sum().backward()
    hw = orc.argmax_first_w_then_h(raw)[0].tolist()
    assert hw == [1, 2]
    assert r.grad[0, 1, 2, 0] == 1.0 and r.grad.sum() == 1.0


def test_project(golden):
    g = golden("g3_project")
    coord = orc.points_in_camera_coords(t(g["z"]), g["rays"]).permute(0, 3, 1, 2)
    assert np.array_equal(coord.numpy(), g["coord"])
    c = coord.clone().requires_grad_(True)
    p = t(g["pose"]).requires_grad_(True)
    grid = orc.coord0TOgrid1(c, p, float(g["cx"]), float(g["cy"]))
    grid.backward(t(g["g_grid"]))
    assert np.array_equal(grid.detach().numpy(), g["grid"])
    assert rel_err(c.grad.numpy(), g["grad_coord"]) < TIGHT
    assert rel_err(p.grad.numpy(), g["grad_pose"]) < 1e-5
    assert np.array_equal(orc.coord0TOgrid1(coord, t(g["pose"])).numpy(), g["grid_native"])
    cr = t(g["coord_r"]).requires_grad_(True)
    tr = t(g["tr"]).requires_grad_(True)
    gr = orc.coord_transform(cr, tr)
    gr.backward(t(g["g_grid_r"]))
    assert np.array_equal(gr.detach().numpy(), g["grid_r"])
    assert rel_err(cr.grad.numpy(), g["grad_coord_r"]) < TIGHT
    assert rel_err(tr.grad.numpy(), g["grad_tr"]) < 1e-5


def test_bilinear(golden):
    g = golden("g4_bilinear")
    for tag in "abc":
        img = t(g[f"{tag}_img"]).requires_grad_(True)
        co = t(g[f"{tag}_coords"]).requires_grad_(True)
        out = orc.bilinear_sampler(img, co)
        out.backward(t(g[f"{tag}_g"]))
        assert np.array_equal(out.detach().numpy(), g[f"{tag}_out"]), tag
        assert rel_err(img.grad.numpy(), g[f"{tag}_grad_img"]) < TIGHT
        assert rel_err(co.grad.numpy(), g[f"{tag}_grad_coords"]) < TIGHT
    # documented quirk: an exact-integer coordinate on the last row/col samples 0
    img = torch.arange(16.0).reshape(1, 4, 4, 1) + 1
    assert float(orc.bilinear_sampler(img, torch.tensor([[[[3.0, 3.0]]]]))) == 0.0


def test_visible_mask(golden):
    g = golden("g5_visible")
    grid = t(g["grid"])
    for m in ("all", "round"):
        assert np.array_equal(orc.gen_visible_mask(grid, m).numpy().astype(np.uint8), g[f"mask_{m}"]), m
        got = np.packbits(orc.gen_visible_mask(t(g["grid_smooth"]), m).numpy().astype(np.uint8).reshape(-1))
        assert np.array_equal(got, g[f"mask_smooth_{m}"]), m
    # wrap-around: x=-1,y=5 on W=8 marks pixel 39; NaN marks nothing
    one = torch.full((1, 16, 8, 2), float("nan"))
    one[0, 0, 0] = torch.tensor([-1.0, 5.0])
    mk = orc.gen_visible_mask(one, "all").reshape(-1)
    assert mk.nonzero().flatten().tolist() == [39]
    with pytest.raises(UnboundLocalError):
        orc.gen_visible_mask(grid, "nearest")


def test_warp_small(golden):
    g = golden("g6_warp_small")
    coord_p = orc.points_in_camera_coords(t(g["z_p"]), g["rays"]).permute(0, 3, 1, 2)
    coord_c = orc.points_in_camera_coords(t(g["z_c"]), g["rays"]).permute(0, 3, 1, 2)
    ip = t(g["Ip"]).requires_grad_(True)
    Ic, gp2c, gc2p = orc.warp_p2c(ip, coord_p, coord_c, t(g["pose"]), float(g["cx"]), float(g["cy"]),
                                  pose_c2p=t(g["pose_c2p"]))
    Ic.backward(t(g["g"]))
    assert np.array_equal(gp2c.numpy(), g["grid_p2c"])
    assert np.array_equal(gc2p.numpy(), g["grid_c2p"])
    assert np.array_equal(orc.gen_visible_mask(gp2c, "all").numpy().astype(np.uint8), g["mask"])
    assert np.array_equal(Ic.detach().numpy(), g["Ic"])
    assert rel_err(ip.grad.numpy(), g["grad_Ip"]) < TIGHT
    frac = float(g["mask"].mean())
    assert 0.2 < frac < 0.999, frac      # the scene really occludes something


def test_warp_native(golden):
    g = golden("g6_warp_native")
    sc = synth.make_scene(1, 800, 1200, int(g["seed"]), pose_scale=1.0)
    assert np.array_equal(sc["pose_p2c"], g["pose"])
    coord_p = orc.z2pointcloud(t(sc["z_p"]))
    coord_c = orc.z2pointcloud(t(sc["z_c"]))
    Ip = np.random.RandomState(int(g["ip_seed"])).uniform(0, 1, (1, 800, 1200, 1)).astype(np.float32)
    Ic, gp2c, gc2p = orc.warp_p2c(t(Ip), coord_p, coord_c, t(sc["pose_p2c"]), pose_c2p=t(g["pose_c2p"]))
    s = int(g["stride"])
    assert np.array_equal(gp2c[:, ::s, ::s].numpy(), g["grid_p2c_sub"])
    assert np.array_equal(gc2p[:, ::s, ::s].numpy(), g["grid_c2p_sub"])
    assert np.array_equal(Ic[:, ::s, ::s].numpy(), g["Ic_sub"])
    mask = orc.gen_visible_mask(gp2c, "all")
    assert np.array_equal(np.packbits(mask.numpy().astype(np.uint8).reshape(-1)), g["mask_bits"])
    assert abs(float(Ic.double().sum()) - float(g["Ic_sum"])) < 1e-6 * float(g["Ic_sum"])


def test_reproj_chain(golden):
    g = golden("g7_reproj")
    zp = t(g["zp"]).requires_grad_(True)
    zc, grid, Ic = orc.reproject_chain(zp, t(g["pose"]), t(g["xy"]), t(g["IpGT"]), float(g["cx"]), float(g["cy"]),
                                       pose_c2p=t(g["pose_c2p"]))
    ((Ic * t(g["g_ic"])).sum() + (zc * t(g["g_zc"])).sum()).backward()
    assert np.array_equal(zc.detach().numpy(), g["zc"])
    assert np.array_equal(grid.detach().numpy(), g["grid"])
    assert np.array_equal(Ic.detach().numpy(), g["Ic"])
    assert rel_err(zp.grad.numpy(), g["grad_zp"]) < TIGHT


def test_reproj_chain_every_gradient_and_crops_per_sample(golden):
    """The chain with every input asking for a gradient (the reference's autograd differentiates the depth, the
    pose -- through linalg.inv and both uses of it --, xy and the pattern), zp_cView_to_zc on its own, and two
    crops of one sample (B_sub = 2: broadcast pose, repeated inverse and pattern)."""
    g = golden("g11_reproj_grads")
    cx, cy = float(g["cx"]), float(g["cy"])
    leaves = [t(g[k]).clone().requires_grad_(True) for k in ("zp", "pose", "xy", "IpGT")]
    zc, _, Ic = orc.reproject_chain(leaves[0], leaves[1], leaves[2], leaves[3], cx, cy)
    ((Ic * t(g["g_ic"])).sum() + (zc * t(g["g_zc"])).sum()).backward()
    assert np.array_equal(zc.detach().numpy(), g["zc"]) and np.array_equal(Ic.detach().numpy(), g["Ic"])
    for leaf, k in zip(leaves, ("grad_zp", "grad_pose", "grad_xy", "grad_IpGT")):
        assert rel_err(leaf.grad.numpy(), g[k]) < TIGHT, k
    leaves = [t(g[k]).clone().requires_grad_(True) for k in ("zp", "pose", "xy")]
    (orc.zp_cView_to_zc(*leaves) * t(g["g_zc"])).sum().backward()
    for leaf, k in zip(leaves, ("zc_only_grad_zp", "zc_only_grad_pose", "zc_only_grad_xy")):
        assert rel_err(leaf.grad.numpy(), g[k]) < TIGHT, k
    leaves = [t(g["zp"]).clone().requires_grad_(True), t(g["pose"][0:1]).clone().requires_grad_(True),
              t(g["xy"]).clone().requires_grad_(True), t(g["IpGT"][0:1]).clone().requires_grad_(True)]
    zc, _, Ic = orc.reproject_chain(*leaves, cx, cy)
    ((Ic * t(g["g_ic"])).sum() + (zc * t(g["g_zc"])).sum()).backward()
    assert np.array_equal(zc.detach().numpy(), g["sub_zc"]) and np.array_equal(Ic.detach().numpy(), g["sub_Ic"])
    for leaf, k in zip(leaves, ("sub_grad_zp", "sub_grad_pose", "sub_grad_xy", "sub_grad_IpGT")):
        assert rel_err(leaf.grad.numpy(), g[k]) < TIGHT, k


def _g9_inputs(g):
    """Scene, pattern, PSFs and the three random draws of the g9 fixture, rebuilt from its seeds."""
    from fc3d import synth
    sc = synth.make_scene(1, 800, 1200, int(g["scene_seed"]), pose_scale=1.0)
    psfs = synth.make_psfs(21, 5, "disc")
    pattern = torch.from_numpy(synth.dot_pattern(800, 1200)).view(1, 800, 1200, 1)
    torch.manual_seed(int(g["noise_seed"]))
    torch.randn(1, 1, 1, 1)            # gen_pattern's unused dot-intensity draw comes first (utils_pytorch.py:25)
    noise = dict(offset=torch.rand(1, 1, 1, 1), gauss=torch.randn(1, 800, 1200, 1), scale=torch.rand(1, 1, 1, 1))
    return sc, psfs, pattern, noise


def check_g9_noise(g, tag, noise):
    # the regenerated draws are the ones the reference consumed
    assert np.array_equal(noise["offset"].numpy(), g[f"{tag}_noise_offset"])
    assert np.array_equal(noise["scale"].numpy(), g[f"{tag}_noise_scale"])
    assert np.array_equal(noise["gauss"].reshape(-1)[:16].numpy(), g[f"{tag}_noise_gauss_head"])
    assert float(noise["gauss"].double().sum()) == float(g[f"{tag}_noise_gauss_sum"])


@pytest.mark.parametrize("tag", ["conv", "noconv"])
def test_callers_match_reference_outputs(golden, tag):
    """The oracle's forward_model / reProj_IpGT against what the reference's own functions returned
    at 800x1200 (train_pytorch.py:79-162, :179-211), bit for bit on the stored lattice."""
    g = golden("g9_callers")
    sc, psfs, pattern, noise = _g9_inputs(g)
    check_g9_noise(g, tag, noise)
    out = orc.forward_model(t(sc["z_p"]), t(sc["z_c"]), t(sc["pose_p2c"]), pattern.expand(-1, -1, -1, 21), psfs, noise,
                            is_conv_psf=(tag == "conv"), pose_c2p=t(g["pose_c2p"]))
    Ic_s, Ic_m, xyz, xy, zc, zp, pose, _, Ic, Ip_coded, Ic_s_full = out
    assert np.array_equal(Ic_s[:, ::4, ::4].numpy(), g[f"{tag}_Ic_scaled_crop_s4"])
    assert np.array_equal(xyz[:, ::4, ::4].numpy(), g[f"{tag}_xyz_cView_crop_s4"])
    assert np.array_equal(Ic[:, ::4, ::4].numpy(), g[f"{tag}_Ic_crop_s4"])
    assert np.array_equal(Ip_coded[:, 300:460, 500:660].numpy(), g[f"{tag}_Ip_coded_patch"])
    assert float(np.abs(g[f"{tag}_Ip_coded_patch"]).max()) > 0.1
    assert np.array_equal(Ic_s_full[:, ::8, ::8].numpy(), g[f"{tag}_Ic_scaled_s8"])
    assert np.array_equal(np.packbits(Ic_m.numpy().astype(np.uint8).reshape(-1)), g[f"{tag}_Ic_mask_crop_bits"])
    for k, v in (("Ic_scaled_crop", Ic_s), ("xyz_cView_crop", xyz), ("Ic_crop", Ic), ("Ip_coded", Ip_coded), ("Ic_scaled", Ic_s_full)):
        assert abs(float(v.double().sum()) - float(g[f"{tag}_{k}_sum"])) <= 1e-9 * abs(float(g[f"{tag}_{k}_sum"])), k
    assert np.array_equal(xy[0, [0, 0, -1, -1], [0, -1, 0, -1]].numpy(), g[f"{tag}_xy_crop_corners"])
    torch.manual_seed(int(g["zhat_seed"]))
    zhat = (xyz[..., 2:3] + 0.003 * torch.randn(1, 512, 512, 1)).requires_grad_(True)
    losses = orc.reProj_IpGT(zhat, pose, xy, Ic, zc, Ic_m, Ip_coded, pose_c2p=t(g["pose_c2p"]))
    losses[0].backward()
    assert np.array_equal(np.array([float(v) for v in losses], np.float32), g[f"{tag}_reproj_losses"])
    assert rel_err(zhat.grad[:, ::4, ::4].numpy(), g[f"{tag}_reproj_grad_s4"]) < TIGHT
    assert abs(float(zhat.grad.double().sum()) - float(g[f"{tag}_reproj_grad_sum"])) <= 1e-6 * float(g[f"{tag}_reproj_grad_abs_sum"])
