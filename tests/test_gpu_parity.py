"""HIP path vs the fixtures captured from the reference and vs the CPU oracle.

Run on the GPU box:  python -m pytest tests -m gpu
Bars: bit-exact for layer indices, gather positions (grids that decide them,
when fed the same matrices) and visibility masks; 1e-4 relative to the tensor's
scale for floating-point results (TOL below)."""
import numpy as np
import pytest
import torch

from conftest import rel_err, rel_err_elem

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import geometry_pytorch as G
    import optics_pytorch as O
    from fc3d import ops, synth
    import fc3d_oracle as orc
    return dict(G=G, O=O, ops=ops, synth=synth, orc=orc)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def full_pattern(p0, L, seed):
    B, H, W, _ = p0.shape
    rest = np.random.RandomState(seed).uniform(0, 1, (B, H, W, L)).astype(np.float32)
    rest[..., 0:1] = p0
    return rest


def patched_K(cx, cy):
    def get_intrinsic(B, device='cuda'):
        K = np.diag([1666.67, 1666.67, 1, 1])
        K[0][2] = cx
        K[1][2] = cy
        return torch.from_numpy(K).float().to(device).unsqueeze(0).repeat(B, 1, 1)
    return get_intrinsic


# --------------------------------------------------------------------------- optics
def test_layer_index_bit_exact(mods, golden):
    g = golden("g1_layers")
    O, ops = mods["O"], mods["ops"]
    z = dev(g["z"])                                  # [1,1,N,1]
    for L in (21, 41, 20, 5):
        pat = torch.ones_like(z)
        _, layer, _, _ = ops.CodePatternFn.apply(pat, z, None, O._layer_grid(L), False)
        got = layer.cpu().numpy().astype(np.int16)
        got[got == 255] = -1
        assert np.array_equal(got.reshape(-1), g[f"layer_L{L}"].reshape(-1)), L


def test_layer_index_every_float_of_a_binade(mods):
    """The 5x5 kernel derives the layer with a reciprocal instruction and re-does ambiguous pixels
    with the exact chain; every float in [0.5, 1) (the working range 0.69..0.95 lies inside) must
    come out as the oracle's layer, for several layer grids."""
    O, ops, synth, orc = mods["O"], mods["ops"], mods["synth"], mods["orc"]
    H, W = 2048, 4096
    bits = np.arange(H * W, dtype=np.uint32) + np.uint32(0x3F000000)      # 0.5 .. 1.0 - ulp
    z = bits.view(np.float32).reshape(1, H, W, 1)
    zt = dev(z)
    pat = torch.ones_like(zt)
    for L in (21, 41, 20):
        psf = dev(synth.make_psfs(L, 5, "disc"))
        _, layer, _, _ = ops.CodePatternFn.apply(pat, zt, psf, O._layer_grid(L), True)
        got = layer.cpu().numpy().astype(np.int16).reshape(-1)
        got[got == 255] = -1
        ref = orc.layer_index(t(z), L).numpy().reshape(-1).astype(np.int16)
        assert np.array_equal(got, ref), (L, int((got != ref).sum()))
    # hostile depths take the same decisions as the exact chain
    odd = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3e38, -0.8, 0.8, 1e-3, 1e3],
                   np.float32)
    zz = np.full((1, 8, 64, 1), 0.8, np.float32)
    zz[0, 3, :odd.size, 0] = odd
    psf = dev(synth.make_psfs(21, 5, "disc"))
    _, layer, _, _ = ops.CodePatternFn.apply(torch.ones(1, 8, 64, 1, device="cuda"), dev(zz), psf, O._layer_grid(21), True)
    got = layer.cpu().numpy().astype(np.int16)
    got[got == 255] = -1
    assert np.array_equal(got[..., 0] if got.ndim == 4 else got, orc.layer_index(t(zz), 21)[..., 0].numpy())


def _cases(g):
    for name in g["cases"]:
        tag, B, H, W, psf, K, conv, pk = str(name).split("|")
        yield tag, int(B[1:]), int(H[1:]), int(W[1:]), psf, int(K[1:]), bool(int(conv[4:])), pk


def test_code_pattern_golden(mods, golden):
    g = golden("g2_code_pattern")
    O, synth = mods["O"], mods["synth"]
    worst = 0.0
    for tag, B, H, W, psf, K, conv, pk in _cases(g):
        O.PSFs_np = synth.make_psfs(21, K, psf)
        pat = dev(full_pattern(g[f"{tag}_p0"], 21, int(g[f"{tag}_patseed"]))).requires_grad_(True)
        out = O.codePattern(pat, dev(g[f"{tag}_z"]), is_conv_psf=conv)
        out.backward(dev(g[f"{tag}_g"]))
        e1 = rel_err(out.detach().cpu().numpy(), g[f"{tag}_out"])
        e2 = rel_err(pat.grad[..., 0:1].cpu().numpy(), g[f"{tag}_grad0"])
        worst = max(worst, e1, e2)
        assert e1 < TOL and e2 < TOL, (tag, psf, K, conv, pk, e1, e2)
        # ... and element by element (where the reference value is not itself at rounding level)
        e3 = rel_err_elem(out.detach().cpu().numpy(), g[f"{tag}_out"])
        e4 = rel_err_elem(pat.grad[..., 0:1].cpu().numpy(), g[f"{tag}_grad0"])
        assert e3 < TOL and e4 < TOL, (tag, psf, K, conv, pk, e3, e4)
        assert float(pat.grad[..., 1:].abs().max()) == 0.0
    print("code_pattern worst rel err", worst)


def test_code_pattern_single_channel_and_expand(mods, golden):
    """The cheap-gradient calling forms give the same numbers as the L-channel form."""
    g = golden("g2_code_pattern")
    O, synth = mods["O"], mods["synth"]
    tag, B, H, W, psf, K, conv, pk = next(c for c in _cases(g) if c[5] == 11)
    O.PSFs_np = synth.make_psfs(21, K, psf)
    z, gg = dev(g[f"{tag}_z"]), dev(g[f"{tag}_g"])
    p0 = dev(g[f"{tag}_p0"])
    ref = dev(g[f"{tag}_out"])
    a = p0.clone().requires_grad_(True)
    o1 = O.codePattern(a, z, n_layers=21)
    o1.backward(gg)
    b = p0.clone().requires_grad_(True)
    o2 = O.codePattern(b.expand(-1, -1, -1, 21), z)
    o2.backward(gg)
    assert torch.equal(o1, o2) and torch.equal(a.grad, b.grad)
    assert rel_err(o1.detach().cpu().numpy(), ref.cpu().numpy()) < TOL
    assert rel_err(a.grad.cpu().numpy(), g[f"{tag}_grad0"]) < TOL


def test_code_pattern_vs_oracle_ragged(mods):
    """Sizes that are not multiples of the tile or of 4, even kernels, L=41."""
    O, synth, orc = mods["O"], mods["synth"], mods["orc"]
    rs = np.random.RandomState(5)
    for (B, H, W, L, K, kind) in ((1, 19, 37, 21, 5, "rand"), (2, 33, 70, 41, 8, "rand"), (1, 70, 130, 41, 23, "disc"),
                                  (1, 17, 66, 21, 33, "gauss"), (3, 16, 64, 21, 1, "rand"), (2, 40, 100, 21, 3, "rand"),
                                  (1, 50, 130, 41, 7, "gauss"), (1, 35, 61, 21, 5, "disc"), (1, 3, 2, 21, 7, "rand")):
        z = rs.uniform(0.69, 0.95, (B, H, W, 1)).astype(np.float32)
        p0 = rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32)
        gg = rs.uniform(-1, 1, (B, H, W, 1)).astype(np.float32)
        psfs = synth.make_psfs(L, K, kind)
        O.PSFs_np = psfs
        a = dev(p0).requires_grad_(True)
        out = O.codePattern(a, dev(z), n_layers=L)
        out.backward(dev(gg))
        pc = t(np.repeat(p0, L, axis=-1)).requires_grad_(True)
        ref = orc.codePattern(pc, t(z), psfs, True)
        ref.backward(t(gg))
        assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TOL, (H, W, K)
        assert rel_err(a.grad.cpu().numpy(), pc.grad[..., 0:1].numpy()) < TOL, (H, W, K)
        li = orc.layer_index(t(z), L)
        _, layer, _, _ = mods["ops"].CodePatternFn.apply(dev(p0), dev(z), None, O._layer_grid(L), False)
        got = layer.cpu().numpy().astype(np.int64)
        got[got == 255] = -1
        assert np.array_equal(got, li[..., 0].numpy())


def test_code_pattern_sweep_tiles_of_both_passes(mods):
    """The large-PSF kernels run the tiles that span many depth layers first and the rest in a second
    pass of the same grid, and stage tiles whose window lies inside the image without clamping: a scene
    with flat parts, a slow ramp, a steep ramp and a step, large enough to hold tiles of every kind
    (border and inner, one layer and many), against the oracle -- layers, image and gradient."""
    O, synth, orc = mods["O"], mods["synth"], mods["orc"]
    rs = np.random.RandomState(77)
    B, H, W, L = 2, 112, 144, 41
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    z0 = np.where(xx < 40, 0.80, np.where(xx < 72, 0.70 + 0.25 * (xx - 40) / 32.0, 0.74 + 0.0004 * yy))
    z0 = np.where((yy > 60) & (xx > 100), 0.93, z0)
    z1 = 0.7 + 0.25 * ((xx * 7 + yy * 3) % 97) / 97.0 * (yy > 48) + 0.1 * (yy <= 48)
    z = np.stack([z0, z1])[..., None].astype(np.float32)
    p0 = rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32)
    gg = rs.uniform(-1, 1, (B, H, W, 1)).astype(np.float32)
    li = orc.layer_index(t(z), L)[..., 0].numpy()
    per_tile = [len(np.unique(li[0, y:y + 16, x:x + 16])) for y in range(0, H, 16) for x in range(0, W, 16)]
    assert min(per_tile) == 1 and max(per_tile) >= 6          # the scene has tiles of both passes
    for K, kind in ((9, "rand"), (23, "disc"), (19, "rand")):
        psfs = synth.make_psfs(L, K, kind)
        O.PSFs_np = psfs
        a = dev(p0).requires_grad_(True)
        out = O.codePattern(a, dev(z), n_layers=L)
        out.backward(dev(gg))
        pc = t(np.repeat(p0, L, axis=-1)).requires_grad_(True)
        ref = orc.codePattern(pc, t(z), psfs, True)
        ref.backward(t(gg))
        assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TOL, K
        assert rel_err(a.grad.cpu().numpy(), pc.grad[..., 0:1].numpy()) < TOL, K
        a2 = dev(p0).requires_grad_(True)
        out2, layer, _, _ = mods["ops"].CodePatternFn.apply(a2, dev(z), O._device_psfs(a2.device), O._layer_grid(L), True)
        got = layer.cpu().numpy().astype(np.int64)
        got[got == 255] = -1
        assert np.array_equal(got, li), K
        (out2 * dev(gg)).sum().backward()
        assert rel_err(a2.grad.cpu().numpy(), pc.grad[..., 0:1].numpy()) < TOL, K


def test_code_pattern_without_blur_vs_oracle(mods):
    """is_conv_psf=False (the mode the reference's training loop runs): ragged and 4-divisible sizes,
    images larger than one wavefront's run, out-of-range depths, a strided channel-0 view."""
    O, orc = mods["O"], mods["orc"]
    rs = np.random.RandomState(15)
    for (B, H, W, L, dense) in ((2, 37, 53, 21, True), (1, 130, 260, 21, True), (3, 64, 96, 41, False), (1, 1, 1, 21, True),
                                (2, 31, 44, 21, False), (1, 2, 2, 21, True)):
        z = rs.uniform(0.67, 0.97, (B, H, W, 1)).astype(np.float32)          # some pixels beyond both ends of the range
        pL = rs.uniform(0, 1, (B, H, W, L)).astype(np.float32)
        gg = rs.uniform(-1, 1, (B, H, W, 1)).astype(np.float32)
        if dense:
            a = dev(np.ascontiguousarray(pL[..., 0:1])).requires_grad_(True)
            out = O.codePattern(a, dev(z), is_conv_psf=False, n_layers=L)
        else:
            a = dev(pL).requires_grad_(True)                                  # channel 0 is a view with stride L
            out = O.codePattern(a, dev(z), is_conv_psf=False)
        out.backward(dev(gg))
        pc = t(pL.copy()).requires_grad_(True)
        ref = orc.codePattern(pc, t(z), None, False)
        ref.backward(t(gg))
        assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TOL, (B, H, W)
        # on the scale of the upstream gradient: at the maximum's pixel g/den and sum(g*out)/den cancel
        # (completely so in a one-pixel image), which leaves rounding noise on either side
        gd = np.abs(a.grad[..., 0:1].cpu().numpy() - pc.grad[..., 0:1].numpy()).max()
        assert gd < TOL * max(float(pc.grad.abs().max()), float(np.abs(gg).max())), (B, H, W)
        if not dense:
            assert float(a.grad[..., 1:].abs().max()) == 0.0


def test_code_pattern_argmax_tie_break(mods):
    """Binary pattern => many pixels share the maximum; the gradient of the max must land
    on the smallest-w-then-smallest-h one, as torch's two-stage max does."""
    O, synth, orc = mods["O"], mods["synth"], mods["orc"]
    B, H, W, L = 2, 32, 48, 21
    z = np.full((B, H, W, 1), 0.8, np.float32)
    p0 = np.broadcast_to(synth.dot_pattern(H, W)[None, :, :, None], (B, H, W, 1)).copy()
    gg = np.random.RandomState(3).uniform(0, 1, (B, H, W, 1)).astype(np.float32)
    a = dev(p0).requires_grad_(True)
    out, _, mx, am = mods["ops"].CodePatternFn.apply(a, dev(z), None, O._layer_grid(L), False)
    out.backward(dev(gg))
    pc = t(np.repeat(p0, L, axis=-1)).requires_grad_(True)
    ref, raw, _ = orc.codePattern(pc, t(z), None, False, return_parts=True)
    ref.backward(t(gg))
    hw = orc.argmax_first_w_then_h(raw.detach())
    assert np.array_equal(am.cpu().numpy(), (hw[:, 0] * W + hw[:, 1]).numpy())
    assert rel_err(a.grad.cpu().numpy(), pc.grad[..., 0:1].numpy()) < TOL


# --------------------------------------------------------------------------- geometry
def test_project_bit_exact(mods, golden, monkeypatch):
    g = golden("g3_project")
    G, ops, orc = mods["G"], mods["ops"], mods["orc"]
    coord = dev(g["coord"])                       # dense [B,4,H,W]
    pose = dev(g["pose"])
    monkeypatch.setattr(G, "get_intrinsic", patched_K(float(g["cx"]), float(g["cy"])))
    c = coord.clone().requires_grad_(True)
    pg = pose.clone().requires_grad_(True)
    grid = G.coord0TOgrid1(c, pg)
    grid.backward(dev(g["g_grid"]))
    assert np.array_equal(grid.detach().cpu().numpy(), g["grid"])
    assert rel_err(c.grad.cpu().numpy(), g["grad_coord"]) < TOL
    assert rel_err_elem(c.grad.cpu().numpy(), g["grad_coord"]) < TOL
    assert rel_err(pg.grad.cpu().numpy(), g["grad_pose"]) < TOL        # K^T (sum over pixels of g_u x coord)
    for b in range(pg.shape[0]):                                       # ... and per sample, entry by entry
        assert np.allclose(pg.grad[b].cpu().numpy(), g["grad_pose"][b], rtol=1e-3, atol=1e-4 * np.abs(g["grad_pose"][b]).max())
    monkeypatch.undo()
    assert np.array_equal(G.coord0TOgrid1(coord, pose).cpu().numpy(), g["grid_native"])
    # permuted [B,H,W,4] view (what z2pointcloud hands over) gives the same bits
    inter = coord.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert not inter.is_contiguous()
    assert np.array_equal(G.coord0TOgrid1(inter, pose).cpu().numpy(), g["grid_native"])
    # odd strides -> generic path
    odd = torch.empty(coord.shape[0], 4, coord.shape[2], coord.shape[3] + 3, device="cuda")[..., 1:-2]
    odd.copy_(coord)
    assert np.array_equal(G.coord0TOgrid1(odd, pose).cpu().numpy(), g["grid_native"])
    # arbitrary matrices through coord_transform
    cr = dev(g["coord_r"]).requires_grad_(True)
    trg = dev(g["tr"]).requires_grad_(True)
    gr = G.coord_transform(cr, trg)
    gr.backward(dev(g["g_grid_r"]))
    assert np.array_equal(gr.detach().cpu().numpy(), g["grid_r"])
    assert rel_err(cr.grad.cpu().numpy(), g["grad_coord_r"]) < TOL
    assert rel_err_elem(cr.grad.cpu().numpy(), g["grad_coord_r"]) < TOL
    assert rel_err(trg.grad.cpu().numpy(), g["grad_tr"]) < TOL
    # only the transform asks for a gradient; a strided view of the points
    trg2 = dev(g["tr"]).requires_grad_(True)
    inter_r = dev(g["coord_r"]).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    G.coord_transform(inter_r, trg2).backward(dev(g["g_grid_r"]))
    assert torch.equal(trg2.grad, trg.grad)
    # tiny image: torch switches to its unfused small-matrix product below 25 pixels
    tiny = dev(g["coord_r"][:, :, :4, :5])
    ref = orc.coord_transform(t(g["coord_r"][:, :, :4, :5]), t(g["tr"]))
    assert np.array_equal(G.coord_transform(tiny, dev(g["tr"])).cpu().numpy(), ref.numpy())


def test_points_and_z2pointcloud(mods, golden):
    g = golden("g3_project")
    G = mods["G"]
    z = dev(g["z"])
    H, W = z.shape[1:3]
    out = G.points_in_camera_coords(z, G.pixel_to_ray_array(W, H))
    assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), g["coord"])
    out2 = G.points_in_camera_coords(z, np.array(g["rays"]))          # unregistered table -> dense upload
    assert torch.equal(out, out2)
    rays_b = dev(np.broadcast_to(g["rays"], (z.shape[0], H, W, 3)).astype(np.float32))
    zz = z.clone().requires_grad_(True)
    out3 = G.points_in_camera_coords(zz, rays_b, allBatch=True)
    assert torch.equal(out, out3.detach())
    gg = torch.rand_like(out3)
    out3.backward(gg)
    ref = (gg[..., :3] * rays_b).sum(-1, keepdim=True)
    assert rel_err(zz.grad.cpu().numpy(), ref.cpu().numpy()) < 1e-6
    with pytest.raises(RuntimeError):
        G.z2pointcloud(z)                                              # 1200x800 table only, like the reference
    g0 = golden("g0_rays")
    assert np.abs(G.pixel_to_ray_array(24, 10) - g0["rays_24x10"]).max() < 1e-15
    assert np.array_equal(G.get_intrinsic(2).cpu().numpy(), g0["intrinsic"])


@pytest.mark.parametrize("mode", ["ordered", "atomic", "atomic+deterministic"])
def test_bilinear_golden(mods, golden, mode):
    """Both image-gradient paths: the gather form without atomics (default) and float-atomic
    accumulation; torch.use_deterministic_algorithms(True) selects the former whatever the mode."""
    g = golden("g4_bilinear")
    G = mods["G"]
    for tag in "abc":
        img = dev(g[f"{tag}_img"]).requires_grad_(True)
        co = dev(g[f"{tag}_coords"]).requires_grad_(True)
        out = G.bilinear_sampler(img, co)
        old = mods["ops"].set_image_gradient_mode(mode.split("+")[0])
        torch.use_deterministic_algorithms(mode.endswith("deterministic"), warn_only=True)
        try:
            out.backward(dev(g[f"{tag}_g"]))
            grads = [img.grad.clone()]
            if mode != "atomic" and img.shape[-1] == 1:   # one channel, fixed summation order: the same bits every time
                img.grad = None
                G.bilinear_sampler(img, co.detach()).backward(dev(g[f"{tag}_g"]))
                assert torch.equal(grads[0], img.grad), tag
        finally:
            torch.use_deterministic_algorithms(False)
            mods["ops"].set_image_gradient_mode(old)
        same = np.array_equal(out.detach().cpu().numpy(), g[f"{tag}_out"])
        assert same, tag
        # A sample a million pixels outside the image adds +-1e12-sized taps that cancel on one
        # border pixel; what is left there depends on summation order in the reference as well,
        # so pixels touched by such samples are not compared.
        keep = np.ones(g[f"{tag}_grad_img"].shape, bool)
        cc = g[f"{tag}_coords"]
        Hs, Ws = keep.shape[1:3]
        far = (np.abs(cc[..., 0]) > Ws + 1000) | (np.abs(cc[..., 1]) > Hs + 1000)
        for b, i, j in zip(*np.nonzero(far)):
            xs = np.clip(np.floor(cc[b, i, j, 0]) + np.array([0, 1]), 0, Ws - 1).astype(int)
            ys = np.clip(np.floor(cc[b, i, j, 1]) + np.array([0, 1]), 0, Hs - 1).astype(int)
            keep[b, ys[:, None], xs[None, :]] = False
        assert (~keep).sum() <= min(4 * int(far.sum()), max(1, keep.size // 50)), (tag, int((~keep).sum()), keep.size)   # a handful of border pixels
        e_img = rel_err(img.grad.cpu().numpy() * keep, g[f"{tag}_grad_img"] * keep)
        e_co = rel_err(co.grad.cpu().numpy(), g[f"{tag}_grad_coords"])
        assert e_img < TOL and e_co < TOL, (tag, e_img, e_co)
        # element by element, for elements above 1 % of the scale: an image-gradient pixel is a signed sum of
        # taps of the scale's size, so a pixel of 1e-3 of the scale is a cancellation to three digits and carries
        # the summation-order noise of its terms (seen: 1.2e-4 there, 2e-5 above 1 %)
        ee_img = rel_err_elem(img.grad.cpu().numpy() * keep, g[f"{tag}_grad_img"] * keep, floor=1e-2)
        ee_co = rel_err_elem(co.grad.cpu().numpy(), g[f"{tag}_grad_coords"])
        assert ee_img < TOL and ee_co < TOL, (tag, ee_img, ee_co)
    img = torch.arange(16.0, device="cuda").reshape(1, 4, 4, 1) + 1
    assert float(G.bilinear_sampler(img, torch.tensor([[[[3.0, 3.0]]]], device="cuda"))) == 0.0


@pytest.mark.parametrize("B,Hs,Ws,Ht,Wt,kind", [
    (2, 37, 53, 37, 53, "smooth"), (1, 20, 70, 33, 41, "random"), (1, 130, 300, 64, 64, "zoom"),
    (3, 5, 3, 5, 3, "smooth"), (1, 1, 1, 1, 1, "smooth"), (2, 64, 200, 64, 200, "jump"), (1, 96, 520, 40, 1030, "smooth"),
    # targets of more than 512 index groups (several scan rounds per workgroup): hits that pile up over the rounds
    # and are worked off once, and -- ten target pixels per source cell -- lists that fill up and are worked off
    # in between
    (1, 300, 500, 700, 900, "smooth"), (1, 64, 128, 640, 1024, "zoom"),
    # the source tiles are 4, 6 or 12 rows tall, by the number of workgroups the source images give: the cases
    # above all take 4; two native-size images take 6, four take 12
    (2, 800, 1200, 800, 1200, "jump"), (4, 800, 1200, 800, 1200, "smooth")])
def test_image_gradient_gather_form(mods, B, Hs, Ws, Ht, Wt, kind):
    """d out / d img of the one-channel sampler in gather form (source tiles look up their samples
    in the tap index) against a float64 scatter_add of the four weighted taps: ragged sizes,
    source and target of different sizes, masks, coordinates outside the image, NaNs, grids that
    are not smooth at all, and the index written by a separate call."""
    ops = mods["ops"]
    rs = np.random.RandomState(B * 1000 + Hs + Wt)
    ys, xs = np.meshgrid(np.arange(Ht), np.arange(Wt), indexing="ij")
    if kind == "random":
        gx, gy = rs.uniform(-3, Ws + 3, (B, Ht, Wt)), rs.uniform(-3, Hs + 3, (B, Ht, Wt))
    else:
        sx, sy = (Ws - 1) / max(Wt - 1, 1), (Hs - 1) / max(Ht - 1, 1)
        th = 0.07
        gx = (np.cos(th) * xs * sx - np.sin(th) * ys * sy + 2.3)[None] + rs.uniform(-0.4, 0.4, (B, Ht, Wt))
        gy = (np.sin(th) * xs * sx + np.cos(th) * ys * sy - 1.7)[None] + rs.uniform(-0.4, 0.4, (B, Ht, Wt))
        if kind == "jump":                                   # a depth discontinuity: half the image lands elsewhere
            gx[:, :, Wt // 2:] += 37.5
            gy[:, Ht // 3:, :] -= 11.25
    grid = np.stack([gx, gy], -1).astype(np.float32)
    if grid.size > 40:
        grid[0, 0, 0] = (np.nan, 1.0)
        grid[0, -1, -1] = (1e9, -1e9)
        grid[0, Ht // 2, Wt // 2] = (float(Ws - 1), float(Hs - 1))     # exact last row / column: zero weight
    g = rs.uniform(-1, 1, (B, Ht, Wt, 1)).astype(np.float32)
    mask = (rs.uniform(0, 1, (B, Ht, Wt)) > 0.2).astype(np.uint8)
    gd, gg, gm = dev(grid), dev(g), dev(mask)

    def reference(use_mask):
        x, y = gd[..., 0], gd[..., 1]
        fx, fy = torch.floor(x), torch.floor(y)
        x0, x1 = fx.clamp(0, Ws - 1), (fx + 1).clamp(0, Ws - 1)
        y0, y1 = fy.clamp(0, Hs - 1), (fy + 1).clamp(0, Hs - 1)
        live = (x0 != x1) & (y0 != y1) & ~torch.isnan(x) & ~torch.isnan(y)
        if use_mask:
            live = live & (gm != 0)
        v = torch.where(live, gg[..., 0], torch.zeros_like(x)).double()
        wx0, wx1, wy0, wy1 = (x1 - x).double(), (x - x0).double(), (y1 - y).double(), (y - y0).double()
        out = torch.zeros(B, Hs * Ws, dtype=torch.float64, device="cuda")
        for xa, ya, w in ((x0, y0, wx0 * wy0), (x0, y1, wx0 * wy1), (x1, y0, wx1 * wy0), (x1, y1, wx1 * wy1)):
            w = torch.where(live, w, torch.zeros_like(w))
            idx = (torch.nan_to_num(ya) * Ws + torch.nan_to_num(xa)).long().reshape(B, -1)
            out.scatter_add_(1, idx, (w * v).reshape(B, -1))
        return out.reshape(B, Hs, Ws, 1)

    old = ops.set_image_gradient_mode("ordered")
    try:
        for use_mask in (False, True):
            m = gm if use_mask else None
            ref = reference(use_mask)
            scale = max(float(ref.abs().max()), 1e-30)
            got = ops._bilinear_bwd_img(gg, gd, m, B, Hs, Ws, 1, Ht, Wt)
            assert float((got.double() - ref).abs().max()) <= 2e-6 * scale, (kind, use_mask)
            assert torch.equal(got, ops._bilinear_bwd_img(gg, gd, m, B, Hs, Ws, 1, Ht, Wt))   # same bits again
            # the index built by its own entry point, then handed over as ready
            idx = ops._tap_index_buffer(B, Hs, Ws, Ht, Wt, gd.device)
            ops._check(ops._lib.fc3d_bilinear_tap_index(ops._stream(), gd.data_ptr(), m.data_ptr() if use_mask else None,
                                                        B, Hs, Ws, Ht, Wt, idx.data_ptr()), "fc3d_bilinear_tap_index")
            assert torch.equal(got, ops._bilinear_bwd_img(gg, gd, m, B, Hs, Ws, 1, Ht, Wt, index=idx))
            ops.set_image_gradient_mode("atomic")
            atom = ops._bilinear_bwd_img(gg, gd, m, B, Hs, Ws, 1, Ht, Wt)
            ops.set_image_gradient_mode("ordered")
            assert float((atom.double() - ref).abs().max()) <= 2e-6 * scale, (kind, use_mask)
    finally:
        ops.set_image_gradient_mode(old)


def test_visible_mask_bit_exact(mods, golden):
    g = golden("g5_visible")
    G = mods["G"]
    for m in ("all", "round"):
        got = G.gen_visible_mask(dev(g["grid"]), m)
        assert got.shape == (2, 16, 8, 1) and got.dtype == torch.float32
        assert np.array_equal(got.cpu().numpy().astype(np.uint8), g[f"mask_{m}"]), m
        got = G.gen_visible_mask(dev(g["grid_smooth"]), m).cpu().numpy().astype(np.uint8)
        assert np.array_equal(np.packbits(got.reshape(-1)), g[f"mask_smooth_{m}"]), m
    with pytest.raises(UnboundLocalError):
        G.gen_visible_mask(dev(g["grid"]), "nearest")


def test_warp_small(mods, golden, monkeypatch):
    g = golden("g6_warp_small")
    G, ops, orc = mods["G"], mods["ops"], mods["orc"]
    monkeypatch.setattr(G, "get_intrinsic", patched_K(float(g["cx"]), float(g["cy"])))
    H, W = g["z_p"].shape[1:3]
    rays = G.pixel_to_ray_array(W, H)
    coord_p = G.points_in_camera_coords(dev(g["z_p"]), rays).permute(0, 3, 1, 2)
    coord_c = G.points_in_camera_coords(dev(g["z_c"]), rays).permute(0, 3, 1, 2)
    ip = dev(g["Ip"]).requires_grad_(True)
    Ic, gp2c, gc2p = G.warp_p2c(ip, coord_p, coord_c, dev(g["pose"]))
    Ic.backward(dev(g["g"]))
    assert np.array_equal(gp2c.cpu().numpy(), g["grid_p2c"])                      # no inverse involved
    assert np.array_equal(G.gen_visible_mask(gp2c, "all").cpu().numpy().astype(np.uint8), g["mask"])
    assert rel_err(gc2p.cpu().numpy(), g["grid_c2p"]) < 1e-5
    flips = int((np.floor(gc2p.cpu().numpy()) != np.floor(g["grid_c2p"])).sum())
    print("gather-index flips caused by the fp64 inverse:", flips, "of", g["grid_c2p"].size)
    assert rel_err(Ic.detach().cpu().numpy(), g["Ic"]) < TOL
    assert rel_err(ip.grad.cpu().numpy(), g["grad_Ip"]) < TOL
    # stage-wise: with the matrices the reference itself used (LAPACK's fp32 inverse differs in
    # the last bit between hosts, so they are part of the fixture) every output is bit-exact
    pose = t(g["pose"])
    K = orc.get_intrinsic(pose.shape[0], float(g["cx"]), float(g["cy"]))
    T_p2c, T_c2p = dev(g["T_p2c"]), dev(g["T_c2p"])
    Ic2, gp2, gc2, mask = ops.WarpP2CFn.apply(dev(g["Ip"]), coord_p, coord_c, T_p2c, T_c2p, None)
    assert np.array_equal(gp2.cpu().numpy(), g["grid_p2c"])
    assert np.array_equal(gc2.cpu().numpy(), g["grid_c2p"])
    assert np.array_equal(mask.cpu().numpy()[..., None], g["mask"])
    assert np.array_equal(Ic2.cpu().numpy(), g["Ic"])
    # pose_transforms: K@pose is bit-exact, the inverse is within an ulp or two
    tf, ti, a = ops.pose_transforms(dev(g["pose"]), K.cuda(), True, True, True)
    assert np.array_equal(tf.cpu().numpy(), T_p2c.cpu().numpy())
    assert rel_err(a.cpu().numpy(), g["pose_c2p"]) < 1e-6
    assert rel_err(ti.cpu().numpy(), g["T_c2p"]) < 1e-6


def test_warp_pose_in_kernel_matches_separate_transforms(mods, golden, monkeypatch):
    """warp_p2c's default path forms K*pose and K*inverse(pose) inside the projection kernel and lets the
    sampling kernel hand the visibility mask back cleared.  Bit for bit what pose_transforms +
    the transform-fed warp give, also on the second and third call (different poses and depths: a
    mask that was not cleared would show), with and without the tap index, and the matrices it
    leaves in memory are pose_transforms' own."""
    g = golden("g6_warp_small")
    G, ops, synth = mods["G"], mods["ops"], mods["synth"]
    monkeypatch.setattr(G, "get_intrinsic", patched_K(float(g["cx"]), float(g["cy"])))
    H, W = g["z_p"].shape[1:3]
    rays = G.pixel_to_ray_array(W, H)
    B = g["z_p"].shape[0]
    K = G.get_intrinsic(B)
    for rep, (seed, Ksel) in enumerate(((11, K), (12, K[0]), (13, K[:1]), (11, K))):
        sc = synth.make_scene(B, H, W, seed=seed, pose_scale=4.0 if rep == 1 else 1.0)
        coord_p = G.points_in_camera_coords(dev(sc["z_p"]), rays).permute(0, 3, 1, 2)
        coord_c = G.points_in_camera_coords(dev(sc["z_c"]), rays).permute(0, 3, 1, 2)
        pose = dev(sc["pose_p2c"])
        ip = dev(g["Ip"]).requires_grad_(rep != 2)
        Ic, gp2c, gc2p = ops.WarpP2CPoseFn.apply(ip, coord_p, coord_c, pose, Ksel)
        tf, ti, _ = ops.pose_transforms(pose, K, want_fwd=True, want_inv=True)
        ip2 = dev(g["Ip"]).requires_grad_(rep != 2)
        Ic2, gp2, gc2, mask2 = ops.WarpP2CFn.apply(ip2, coord_p, coord_c, tf, ti, None)
        assert torch.equal(gp2c, gp2) and torch.equal(gc2p, gc2) and torch.equal(Ic, Ic2), rep
        if rep != 2:
            Ic.backward(dev(g["g"]))
            Ic2.backward(dev(g["g"]))
            assert torch.equal(ip.grad, ip2.grad), rep
    # a size whose pixel count is not a multiple of 16 (the mask is handed back cleared byte by byte) and that
    # leaves the last workgroup partly empty; strided (planar) coordinates
    Hr, Wr = 37, 53
    rays_r = G.pixel_to_ray_array(Wr, Hr)
    sc = synth.make_scene(2, Hr, Wr, seed=17)
    Kr = dev(np.diag([1666.67, 1666.67, 1, 1]).astype(np.float32))
    Kr[0, 2], Kr[1, 2] = Wr / 2, Hr / 2
    cp = G.points_in_camera_coords(dev(sc["z_p"]), rays_r).permute(0, 3, 1, 2).contiguous()
    cc = G.points_in_camera_coords(dev(sc["z_c"]), rays_r).permute(0, 3, 1, 2)
    ipr = dev(np.random.RandomState(3).uniform(0, 1, (2, Hr, Wr, 1)).astype(np.float32))
    for rep in range(2):
        a = ops.WarpP2CPoseFn.apply(ipr, cp, cc, dev(sc["pose_p2c"]), Kr)
        tf, ti, _ = ops.pose_transforms(dev(sc["pose_p2c"]), Kr, want_fwd=True, want_inv=True)
        b2 = ops.WarpP2CFn.apply(ipr, cp, cc, tf, ti, None)
        assert all(torch.equal(x, y) for x, y in zip(a, b2[:3])), rep
    # the two halves issued apart, the first on another stream: what warp_p2c gives, gradient included
    side = torch.cuda.Stream()
    for rep in range(2):
        ip1 = ipr.clone().requires_grad_(True)
        ip2 = ipr.clone().requires_grad_(True)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            begun = G.warp_p2c_begin(cp, dev(sc["pose_p2c"]), Kr)
        torch.cuda.current_stream().wait_stream(side)
        a = G.warp_p2c_finish(ip1, cc, begun)
        b3 = G._warp_p2c(ip2, cp, cc, dev(sc["pose_p2c"]), Kr)
        assert all(torch.equal(x, y) for x, y in zip(a, b3)), rep
        up = torch.ones_like(a[0])
        a[0].backward(up)
        b3[0].backward(up)
        assert torch.equal(ip1.grad, ip2.grad)
        with pytest.raises(RuntimeError):
            G.warp_p2c_finish(ip1, cc, begun)       # a first half is good for one second half
        if rep == 0:
            G.warp_p2c_begin(cp, dev(sc["pose_p2c"]), Kr)   # never finished: its marked mask must not come back
    for m in ops._zero_masks.values():
        assert int(m.count_nonzero()) == 0          # every pooled mask is back to zeros
    # the matrices written on the side
    lib = ops._lib
    sc = synth.make_scene(B, H, W, seed=21)
    coord_p = G.points_in_camera_coords(dev(sc["z_p"]), rays).permute(0, 3, 1, 2).contiguous()
    pose = dev(sc["pose_p2c"])
    out = [torch.empty((B, H, W, c), device="cuda") for c in (1, 2, 2)]
    mats = torch.full((3, B, 4, 4), float("nan"), device="cuda")
    mask = torch.zeros((B, H, W), dtype=torch.uint8, device="cuda")
    st = coord_p.stride()
    ops._check(lib.fc3d_warp_p2c_pose_fwd(ops._stream(), dev(g["Ip"]).data_ptr(), coord_p.data_ptr(), *st,
                                          coord_p.data_ptr(), *st, pose.data_ptr(), K.data_ptr(), 16, B, H, W,
                                          out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), mask.data_ptr(),
                                          1, 0, None, mats[0].data_ptr(), mats[1].data_ptr(), mats[2].data_ptr()),
               "fc3d_warp_p2c_pose_fwd")
    tf, ti, a = ops.pose_transforms(pose, K, True, True, True)
    assert torch.equal(mats[0], tf) and torch.equal(mats[1], ti) and torch.equal(mats[2], a)
    assert int(mask.count_nonzero()) > 0            # mask_rezero = 0: the mask stays available


def test_warp_reads_the_depth_plane_behind_noted_points(mods, monkeypatch):
    """Scene points made by points_in_camera_coords carry a note of the depth map and rays they came from;
    warp_p2c then reads 4 bytes of depth per pixel instead of the 16-byte point and forms the point as it was
    stored.  Every output and the pattern gradient equal the stored-points form bit for bit; the note is dropped
    as soon as the points or the depth map were written in place, and for anything but the plain permuted view."""
    G, ops, synth = mods["G"], mods["ops"], mods["synth"]
    B, H, W = 3, 70, 93
    monkeypatch.setattr(G, "get_intrinsic", patched_K(W / 2, H / 2))
    sc = synth.make_scene(B, H, W, seed=41, theta_deg=12.0, t_max=(0.1, 0.1, 0.04))
    rays = G.pixel_to_ray_array(W, H)
    z_p, z_c, pose = dev(sc["z_p"]), dev(sc["z_c"]), dev(sc["pose_p2c"])
    rs = np.random.RandomState(42)
    ip0 = rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32)
    gg = dev(rs.uniform(-1, 1, (B, H, W, 1)).astype(np.float32))

    def run(cp, cc):
        ip = dev(ip0).requires_grad_(True)
        Ic, gp2c, gc2p = G.warp_p2c(ip, cp, cc, pose)
        Ic.backward(gg)
        return (Ic.detach(), gp2c, gc2p, ip.grad), ops.WarpP2CPoseFn.last_form

    def points(z):
        return G.points_in_camera_coords(z, rays).permute(0, 3, 1, 2)

    cp, cc = points(z_p), points(z_c)
    got, form = run(cp, cc)
    assert form == "depth"
    old = ops.set_points_from_depth(False)
    try:
        want, form = run(cp, cc)
        assert form == "points"
    finally:
        ops.set_points_from_depth(old)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # copies and other views carry no note
    assert run(cp.clone(), cc)[1] == "points"
    assert run(cp.contiguous(), cc.contiguous())[1] == "points"
    # the points written in place: their values count, not the depth map's
    cp2 = points(z_p)
    cp2._base[..., 2] += 0.01
    got2, form = run(cp2, cc)
    assert form == "points"
    want2, _ = run(cp2.clone(), cc.clone())
    assert all(torch.equal(a, b) for a, b in zip(got2, want2)) and not torch.equal(got2[1], got[1])
    # the depth map written in place after the points were made: the points still say what they said
    z2 = z_p.clone()
    cp3 = points(z2)
    z2 += 0.05
    got3, form = run(cp3, cc)
    assert form == "points"
    assert all(torch.equal(a, b) for a, b in zip(got3, got))
    # a stream capture cannot re-check the note at replay: it reads the stored points unless its owner opts in
    Kc = G.get_intrinsic(B)                                         # (no host-to-device copy inside the capture)
    monkeypatch.setattr(G, "get_intrinsic", lambda B_, device="cuda": Kc)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    ipc = dev(ip0)
    with torch.cuda.stream(side):
        G.warp_p2c(ipc, cp, cc, pose)
    torch.cuda.synchronize()
    for allow, expect in ((False, "points"), (True, "depth")):
        old = ops.allow_points_from_depth_in_capture(allow)
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side, capture_error_mode="thread_local"):
                Icg, _, gcg = G.warp_p2c(ipc, cp, cc, pose)
            assert ops.WarpP2CPoseFn.last_form == expect
        finally:
            ops.allow_points_from_depth_in_capture(old)
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(Icg, got[0]) and torch.equal(gcg, got[2])


def test_two_stream_warp_loop_needs_only_the_forward_wait(mods):
    """An eager loop that issues the first half of warp_p2c on a side stream and orders ONLY
    main.wait_stream(side) before the second half, as the docstring asks: the side stream runs ahead and would
    splat the next scene into the pooled mask while the main stream's sampling kernel still reads and clears
    it -- unless the buffer's next checkout waits for that kernel (the event stored with it).  Every iteration
    must give what the one-call warp_p2c gives for its scene."""
    G, ops, synth = mods["G"], mods["ops"], mods["synth"]
    B, H, W = 8, 512, 512
    rays = G.pixel_to_ray_array(W, H)
    Kmat = dev(np.diag([1666.67, 1666.67, 1, 1]).astype(np.float32))
    Kmat[0, 2], Kmat[1, 2] = W / 2, H / 2
    scenes = []
    for seed in (41, 42, 43):
        sc = synth.make_scene(B, H, W, seed=seed, theta_deg=12.0, t_max=(0.1, 0.1, 0.03))   # strong occlusion: masks differ a lot
        cp = G.points_in_camera_coords(dev(sc["z_p"]), rays).permute(0, 3, 1, 2)
        cc = G.points_in_camera_coords(dev(sc["z_c"]), rays).permute(0, 3, 1, 2)
        scenes.append((cp, cc, dev(sc["pose_p2c"])))
    ip = dev(np.random.RandomState(9).uniform(0.2, 1, (B, H, W, 1)).astype(np.float32))
    want = [tuple(t.clone() for t in G._warp_p2c(ip, cp, cc, pose, Kmat)) for cp, cc, pose in scenes]
    assert not torch.equal(want[0][0], want[1][0])
    torch.cuda.synchronize()
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    side.wait_stream(main)                      # once, for the inputs made above
    busy = torch.rand(4096, 4096, device="cuda")
    got = []
    for it in range(12):
        cp, cc, pose = scenes[it % 3]
        with torch.cuda.stream(side):
            begun = G.warp_p2c_begin(cp, pose, Kmat)       # runs ahead of the main stream
        busy = busy * 1.0001                               # the main stream is behind: work queued in front of the second half
        main.wait_stream(side)
        got.append(G.warp_p2c_finish(ip, cc, begun))
    torch.cuda.synchronize()
    for it, g3 in enumerate(got):
        for a, b in zip(g3, want[it % 3]):
            assert torch.equal(a, b), it
    for m in ops._zero_masks.values():
        assert int(m.count_nonzero()) == 0


@pytest.mark.parametrize("K", [3, 5, 7, 9, 23, 0])
def test_backward_hands_sum_from_sampler_to_code_pattern(mods, K):
    """codePattern -> warp_p2c backward: the sampler's image-gradient kernel hands sum(g * Ip_coded) to
    codePattern's backward (one launch, `out` not read, the maximum's share applied at its source pixel).
    Same gradient as the ordinary two-launch backward to 1e-6 of the scale and element by element, the
    hand-off really happens, and it is refused when the gradient was touched on the way."""
    G, O, ops, synth = mods["G"], mods["O"], mods["ops"], mods["synth"]
    B, H, W, L = 3, 150, 276, 21                       # ragged: tiles and strips that hang over the image
    sc = synth.make_scene(B, H, W, seed=31)
    conv = K > 0                                       # K = 0: is_conv_psf=False, the streaming kernels
    O.PSFs_np = synth.make_psfs(L, K if conv else 5, "disc")
    Kmat = dev(np.diag([1666.67, 1666.67, 1, 1]).astype(np.float32))
    Kmat[0, 2], Kmat[1, 2] = W / 2, H / 2
    rays = G.pixel_to_ray_array(W, H)
    coord_p = G.points_in_camera_coords(dev(sc["z_p"]), rays).permute(0, 3, 1, 2)
    coord_c = G.points_in_camera_coords(dev(sc["z_c"]), rays).permute(0, 3, 1, 2)
    p0 = synth.make_pattern(sc["z_p"], 5, "uniform")
    up = dev(np.random.RandomState(6).uniform(0.0, 1.0, (B, H, W, 1)).astype(np.float32))
    res = {}
    calls = {"dot": 0, "plain": 0}

    class Spy:
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            if name == "fc3d_code_pattern_bwd_dot":
                calls["dot"] += 1
            if name == "fc3d_code_pattern_bwd":
                calls["plain"] += 1
            return getattr(self._lib, name)
    lib = ops._lib
    ops._lib = Spy(lib)
    stats0 = dict(ops.handover_stats)
    try:
        for mode in ("handed", "touched", "no_offer"):
            pat = dev(p0).requires_grad_(True)
            Ip = O.codePattern(pat, dev(sc["z_p"]), is_conv_psf=conv, n_layers=L)
            if mode == "touched":
                Ip.register_hook(lambda g: g.mul_(1.0))          # in place: same tensor, new version
            if mode == "no_offer":
                Ic = G.bilinear_sampler(Ip, G._coord0TOgrid1(coord_c, torch.linalg.inv(dev(sc["pose_p2c"])), Kmat))
            else:
                Ic, _, _ = G._warp_p2c(Ip, coord_p, coord_c, dev(sc["pose_p2c"]), Kmat)
            before = dict(calls)
            Ic.backward(up)
            used = {k: calls[k] - before[k] for k in calls}
            res[mode] = (pat.grad.clone(), used)
    finally:
        ops._lib = lib
    assert res["handed"][1] == {"dot": 1, "plain": 0}
    assert res["touched"][1] == {"dot": 0, "plain": 1} and res["no_offer"][1] == {"dot": 0, "plain": 1}
    # the hand-over keeps count of itself: two offers (handed, touched), one taken, one refused, none lost
    d = {k: ops.handover_stats[k] - stats0[k] for k in stats0}
    assert d == {"offered": 2, "taken": 1, "refused": 1, "lost": 0}, d
    # a second consumer of Ip_coded: autograd sums the two gradients into a new tensor, the offer cannot arrive;
    # that is counted as lost and reported once (the same would show if PyTorch stopped passing the tensor on)
    import warnings
    pat = dev(p0).requires_grad_(True)
    Ip = O.codePattern(pat, dev(sc["z_p"]), is_conv_psf=conv, n_layers=L)
    Ic, _, _ = G._warp_p2c(Ip, coord_p, coord_c, dev(sc["pose_p2c"]), Kmat)
    ops._warned_lost = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ((Ic * up).sum() + 0.5 * Ip.sum()).backward()
    assert ops.handover_stats["lost"] - stats0["lost"] == 1
    assert any("did not reach" in str(w.message) for w in rec)
    both = pat.grad.clone()
    pat2 = dev(p0).requires_grad_(True)
    Ip2 = O.codePattern(pat2, dev(sc["z_p"]), is_conv_psf=conv, n_layers=L)
    (0.5 * Ip2.sum()).backward()
    assert rel_err((both - pat2.grad).cpu().numpy(), res["handed"][0].cpu().numpy()) < 1e-4
    a, b = res["handed"][0].cpu().numpy(), res["touched"][0].cpu().numpy()
    assert float(np.abs(b).max()) > 0
    assert rel_err(a, b) < 1e-6 and rel_err_elem(a, b) < 1e-4


def test_warp_differentiable_geometry_path(mods, golden, monkeypatch):
    """warp_p2c with coords that require grad takes the composed-operator path; same numbers."""
    g = golden("g6_warp_small")
    G = mods["G"]
    monkeypatch.setattr(G, "get_intrinsic", patched_K(float(g["cx"]), float(g["cy"])))
    H, W = g["z_p"].shape[1:3]
    rays = G.pixel_to_ray_array(W, H)
    coord_p = G.points_in_camera_coords(dev(g["z_p"]), rays).permute(0, 3, 1, 2)
    zc = dev(g["z_c"]).requires_grad_(True)
    coord_c = G.points_in_camera_coords(zc, rays).permute(0, 3, 1, 2)
    ip = dev(g["Ip"]).requires_grad_(True)
    Ic, gp2c, gc2p = G.warp_p2c(ip, coord_p, coord_c, dev(g["pose"]))
    Ic.backward(dev(g["g"]))
    assert rel_err(Ic.detach().cpu().numpy(), g["Ic"]) < TOL
    assert rel_err(ip.grad.cpu().numpy(), g["grad_Ip"]) < TOL
    assert zc.grad is not None and torch.isfinite(zc.grad).all()
    # the pose as a differentiable input (through K @ pose and K @ inverse(pose)) against the oracle's
    # autograd on the CPU; the image is smooth here so that the bilinear weights' gradient is well defined
    orc = mods["orc"]
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    smooth = (0.5 + 0.3 * np.sin(xs / 7.0) * np.cos(ys / 5.0)).astype(np.float32)[None, :, :, None].repeat(g["Ip"].shape[0], 0)
    pose = dev(g["pose"]).requires_grad_(True)
    Ic2, _, _ = G.warp_p2c(dev(smooth), coord_p, coord_c.detach(), pose)
    Ic2.backward(dev(g["g"]))
    pr = t(g["pose"]).requires_grad_(True)
    cp, cc = t(coord_p.cpu().numpy()), t(coord_c.detach().cpu().numpy())
    Ic_r, _, _ = orc.warp_p2c(t(smooth), cp, cc, pr, float(g["cx"]), float(g["cy"]))
    Ic_r.backward(t(g["g"]))
    assert rel_err(Ic2.detach().cpu().numpy(), Ic_r.detach().numpy()) < TOL
    assert rel_err(pose.grad.cpu().numpy(), pr.grad.numpy()) < 2e-3     # sums of ~6000 bilinear-weight gradients each


def test_warp_native_800x1200(mods, golden):
    g = golden("g6_warp_native")
    G, synth = mods["G"], mods["synth"]
    sc = synth.make_scene(1, 800, 1200, int(g["seed"]), pose_scale=1.0)
    coord_p = G.z2pointcloud(dev(sc["z_p"]))
    coord_c = G.z2pointcloud(dev(sc["z_c"]))
    assert coord_p.shape == (1, 4, 800, 1200)
    Ip = np.random.RandomState(int(g["ip_seed"])).uniform(0, 1, (1, 800, 1200, 1)).astype(np.float32)
    Ic, gp2c, gc2p = G.warp_p2c(dev(Ip), coord_p, coord_c, dev(sc["pose_p2c"]))
    s = int(g["stride"])
    assert np.array_equal(gp2c[:, ::s, ::s].cpu().numpy(), g["grid_p2c_sub"])
    mask = G.gen_visible_mask(gp2c, "all").cpu().numpy().astype(np.uint8)
    assert np.array_equal(np.packbits(mask.reshape(-1)), g["mask_bits"])
    assert rel_err(gc2p[:, ::s, ::s].cpu().numpy(), g["grid_c2p_sub"]) < 1e-6
    # Ip is white noise here and one fp32 ulp of a coordinate near 1000 is 1.2e-4 pixel, so a
    # last-bit difference in inverse(pose) moves single samples by up to ~2e-4: looser bar for
    # this end-to-end check, bit-exact bar below once the reference's own matrices are used.
    assert rel_err(Ic[:, ::s, ::s].cpu().numpy(), g["Ic_sub"]) < 1e-3
    assert abs(float(Ic.double().sum()) - float(g["Ic_sum"])) < 1e-5 * float(g["Ic_sum"])
    Ic2, gp2, gc2, mk = mods["ops"].WarpP2CFn.apply(dev(Ip), coord_p, coord_c, dev(g["T_p2c"]), dev(g["T_c2p"]), None)
    assert np.array_equal(gc2[:, ::s, ::s].cpu().numpy(), g["grid_c2p_sub"])
    assert np.array_equal(Ic2[:, ::s, ::s].cpu().numpy(), g["Ic_sub"])
    assert np.array_equal(np.packbits(mk.cpu().numpy().reshape(-1)), g["mask_bits"])


def test_reproj_chain(mods, golden, monkeypatch):
    g = golden("g7_reproj")
    G = mods["G"]
    monkeypatch.setattr(G, "get_intrinsic", patched_K(float(g["cx"]), float(g["cy"])))
    zp = dev(g["zp"]).requires_grad_(True)
    xy, pose = dev(g["xy"]), dev(g["pose"])
    zc = G.zp_cView_to_zc(zp, pose, xy)
    ray = torch.cat([xy, torch.ones_like(zc)], dim=-1)
    coord_c = G.points_in_camera_coords(zc, ray, allBatch=True).permute(0, 3, 1, 2)
    tinv = dev(g["pose_c2p"])                                          # the reference's own inverse
    grid = G.coord0TOgrid1(coord_c, tinv)
    Ic = G.bilinear_sampler(dev(g["IpGT"]), grid)
    ((Ic * dev(g["g_ic"])).sum() + (zc * dev(g["g_zc"])).sum()).backward()
    assert rel_err(zc.detach().cpu().numpy(), g["zc"]) < 1e-6
    assert rel_err(grid.detach().cpu().numpy(), g["grid"]) < 1e-5
    assert rel_err(Ic.detach().cpu().numpy(), g["Ic"]) < TOL
    assert rel_err(zp.grad.cpu().numpy(), g["grad_zp"]) < TOL
    # element by element where the value is above 3 % of the scale: this gradient is a sum of two terms of the
    # scale's size (through zc and through the sampled pattern), so a small element is their cancellation and
    # carries the rounding noise of the large ones (measured: 5.8e-5 above 3 %, 3.7e-4 at 1e-3 of the scale)
    assert rel_err_elem(zp.grad.cpu().numpy(), g["grad_zp"], floor=3e-2) < TOL


def test_reproject_fused(mods, golden, monkeypatch):
    """The one-kernel re-projection chain: identical forward results to composing the operators,
    the fixture's values and gradient, and the right gradient when only one output is used."""
    g = golden("g7_reproj")
    G, ops = mods["G"], mods["ops"]
    monkeypatch.setattr(G, "get_intrinsic", patched_K(float(g["cx"]), float(g["cy"])))
    xy, pose, img = dev(g["xy"]), dev(g["pose"]), dev(g["IpGT"])
    B = xy.shape[0]
    inv = dev(g["pose_c2p"])                                           # the reference's own inverse
    K = G.get_intrinsic(B)
    T, _, _ = ops.pose_transforms(inv, K, want_fwd=True)               # K @ inverse(pose)
    # composed
    zp1 = dev(g["zp"]).requires_grad_(True)
    zc1 = G.zp_cView_to_zc(zp1, pose, xy)
    ray = torch.cat([xy, torch.ones_like(zc1)], dim=-1)
    grid1 = G.coord0TOgrid1(G.points_in_camera_coords(zc1, ray, allBatch=True).permute(0, 3, 1, 2), inv)
    Ic1 = G.bilinear_sampler(img, grid1)
    # fused, fed the same inverse
    zp2 = dev(g["zp"]).requires_grad_(True)
    zc2, Ic2 = ops.ReprojectFn.apply(zp2, xy, inv, T, img)
    assert rel_err(zc2.detach().cpu().numpy(), g["zc"]) < 1e-6 and rel_err(Ic2.detach().cpu().numpy(), g["Ic"]) < TOL
    # zc through the library's own fp64 inverse differs from the fixture's LAPACK inverse in the last bit at most
    assert rel_err(zc2.detach().cpu().numpy(), zc1.detach().cpu().numpy()) < 1e-6
    assert rel_err(Ic2.detach().cpu().numpy(), Ic1.detach().cpu().numpy()) < TOL
    ((Ic2 * dev(g["g_ic"])).sum() + (zc2 * dev(g["g_zc"])).sum()).backward()
    assert rel_err(zp2.grad.cpu().numpy(), g["grad_zp"]) < TOL
    assert rel_err_elem(zp2.grad.cpu().numpy(), g["grad_zp"], floor=3e-2) < TOL       # see test_reproj_chain
    for use in ("zc", "Ic"):
        a = dev(g["zp"]).requires_grad_(True)
        b = dev(g["zp"]).requires_grad_(True)
        zca, Ica = ops.ReprojectFn.apply(a, xy, inv, T, img)
        zcb = G.zp_cView_to_zc(b, pose, xy)
        rayb = torch.cat([xy, torch.ones_like(zcb)], dim=-1)
        Icb = G.bilinear_sampler(img, G.coord0TOgrid1(
            G.points_in_camera_coords(zcb, rayb, allBatch=True).permute(0, 3, 1, 2), inv))
        if use == "zc":
            (zca * dev(g["g_zc"])).sum().backward()
            (zcb * dev(g["g_zc"])).sum().backward()
        else:
            (Ica * dev(g["g_ic"])).sum().backward()
            (Icb * dev(g["g_ic"])).sum().backward()
        assert rel_err(a.grad.cpu().numpy(), b.grad.cpu().numpy()) < TOL, use
    # same inverse for both paths => identical bits forward
    from fc3d import capture
    zp3 = dev(g["zp"])
    zf, If = capture.reproject(zp3, pose, xy, img, fused=True)
    zs, Is = capture.reproject(zp3, pose, xy, img, fused=False)
    assert torch.equal(zf, zs) and torch.equal(If, Is)


def test_reproj_chain_every_gradient_and_crops_per_sample(mods, golden, monkeypatch):
    """What the reference's autograd gives for the re-projection chain when everything asks for a gradient --
    predicted depth, pose (through the inverse and both of its uses), xy, pattern --, for zp_cView_to_zc on its own,
    and for two crops per sample (train_pytorch.py:189-202 with B_sub = 2); the one-kernel chain gives the same
    forward values for the crops."""
    from fc3d import capture
    g = golden("g11_reproj_grads")
    G, ops = mods["G"], mods["ops"]
    cx, cy = float(g["cx"]), float(g["cy"])
    monkeypatch.setattr(G, "get_intrinsic", patched_K(cx, cy))
    tol_pose = 2e-3          # sums of ~3000 per-pixel terms of mixed sign each, against LAPACK's fp32 inverse

    def check(leaves, keys, zc, Ic, pre):
        ((Ic * dev(g["g_ic"])).sum() + (zc * dev(g["g_zc"])).sum()).backward()
        assert rel_err(zc.detach().cpu().numpy(), g[pre + "zc"]) < 1e-6
        assert rel_err(Ic.detach().cpu().numpy(), g[pre + "Ic"]) < TOL
        for leaf, k in zip(leaves, keys):
            e = rel_err(leaf.grad.cpu().numpy(), g[pre + k])
            assert e < (tol_pose if k == "grad_pose" else TOL), (pre + k, e)

    leaves = [dev(g[k]).clone().requires_grad_(True) for k in ("zp", "pose", "xy", "IpGT")]
    zc, Ic = capture.reproject(*leaves, intrinsics=(cx, cy))
    check(leaves, ("grad_zp", "grad_pose", "grad_xy", "grad_IpGT"), zc, Ic, "")
    # zp_cView_to_zc alone, through the drop-in module
    leaves = [dev(g[k]).clone().requires_grad_(True) for k in ("zp", "pose", "xy")]
    (G.zp_cView_to_zc(*leaves) * dev(g["g_zc"])).sum().backward()
    for leaf, k in zip(leaves, ("zc_only_grad_zp", "zc_only_grad_pose", "zc_only_grad_xy")):
        e = rel_err(leaf.grad.cpu().numpy(), g[k])
        assert e < (tol_pose if k.endswith("pose") else TOL), (k, e)
    # two crops of one sample
    leaves = [dev(g["zp"]).clone().requires_grad_(True), dev(g["pose"][0:1]).clone().requires_grad_(True),
              dev(g["xy"]).clone().requires_grad_(True), dev(g["IpGT"][0:1]).clone().requires_grad_(True)]
    zc, Ic = capture.reproject(*leaves, intrinsics=(cx, cy))
    check(leaves, ("grad_zp", "grad_pose", "grad_xy", "grad_IpGT"), zc, Ic, "sub_")
    zp2 = dev(g["zp"]).clone().requires_grad_(True)
    zc_f, Ic_f = capture.reproject(zp2, dev(g["pose"][0:1]), dev(g["xy"]), dev(g["IpGT"][0:1]), intrinsics=(cx, cy))     # one kernel
    assert torch.equal(zc_f, zc.detach()) and torch.equal(Ic_f, Ic.detach())
    ((Ic_f * dev(g["g_ic"])).sum() + (zc_f * dev(g["g_zc"])).sum()).backward()
    assert rel_err(zp2.grad.cpu().numpy(), g["sub_grad_zp"]) < TOL
    # the per-sample ray map's gradient on its own, against the tensor expression
    z = dev(g["zp"]).clone().requires_grad_(True)
    ray = torch.cat([dev(g["xy"]), torch.ones_like(z)], dim=-1).requires_grad_(True)
    gg = torch.rand(z.shape[0], z.shape[1], z.shape[2], 4, device="cuda")
    G.points_in_camera_coords(z, ray, allBatch=True).backward(gg)
    assert torch.equal(ray.grad, gg[..., :3] * z.detach())


def test_empty_batch(mods):
    """B = 0 gives empty results of the right shapes (the reference's tensor ops do the same)."""
    G, O = mods["G"], mods["O"]
    H, W = 8, 12
    z = torch.zeros(0, H, W, 1, device="cuda")
    assert tuple(O.codePattern(torch.zeros(0, H, W, 21, device="cuda"), z).shape) == (0, H, W, 1)
    pts = G.points_in_camera_coords(z, G.pixel_to_ray_array(W, H))
    assert tuple(pts.shape) == (0, H, W, 4)
    coord = pts.permute(0, 3, 1, 2)
    pose = torch.zeros(0, 4, 4, device="cuda")
    Ic, g1, g2 = G.warp_p2c(z, coord, coord, pose)
    assert tuple(Ic.shape) == (0, H, W, 1) and tuple(g1.shape) == (0, H, W, 2) and tuple(g2.shape) == (0, H, W, 2)
    assert tuple(G.coord0TOgrid1(coord, pose).shape) == (0, H, W, 2)
    assert tuple(G.bilinear_sampler(torch.zeros(0, H, W, 3, device="cuda"), g1).shape) == (0, H, W, 3)
    assert tuple(G.gen_visible_mask(g1, "all").shape) == (0, H, W, 1)
    assert tuple(G.zp_cView_to_zc(z, pose, g1).shape) == (0, H, W, 1)


def test_errors_are_loud(mods):
    G, O = mods["G"], mods["O"]
    with pytest.raises(RuntimeError):
        O.codePattern(torch.zeros(1, 8, 8, 21), torch.ones(1, 8, 8, 1))          # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        G.bilinear_sampler(torch.zeros(1, 4, 4, 1), torch.zeros(1, 4, 4, 2))
    with pytest.raises(TypeError):
        G.bilinear_sampler(torch.zeros(1, 4, 4, 1, device="cuda").double(), torch.zeros(1, 4, 4, 2, device="cuda"))
    with pytest.raises(RuntimeError):
        O.codePattern(torch.zeros(1, 8, 8, 7, device="cuda"), torch.ones(1, 8, 8, 1, device="cuda"))  # 7 != 21 PSFs
