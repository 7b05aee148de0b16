"""GPU: each HIP kernel family through the C ABI against a plain PyTorch fp32 reference of the same op
(operands rounded to the build's MFMA operand type first -- bf16 for libdfengine.so, fp16 for libdfengine_f16.so --
so the only difference is fp32 accumulation order): tolerance 2e-3 relative.  Every test runs on both builds."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from helpers import rnd, rel_l2

pytestmark = pytest.mark.gpu


def _eng():
    from diff_foley_amd import engine as E
    return E


PREC = "bf16"


@pytest.fixture(params=["bf16", "fp16"], autouse=True)
def prec(request):
    global PREC
    PREC = request.param
    yield PREC
    PREC = "bf16"


def odt():
    return _eng().OPERAND_DTYPE[PREC]


def bf(t):
    return t.to(odt())


def ptr(t):
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


TILES = [0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 13, 14, 18, 19, 20, 26, 27, 28, 29]      # 18-20, 26-29: producer-specialised blocks
HALO = (5, 6, 7, 15, 16, 17, 23, 24, 25)       # 23-25: producer-specialised (4 consumer + 4 producer wavefronts)
_HALO_GEO = {5: (128, 64, 256, 4), 6: (256, 64, 512, 4), 7: (128, 128, 256, 4), 15: (128, 64, 256, 8), 16: (256, 64, 512, 8),
             17: (192, 64, 256, 4), 23: (192, 64, 256, 4), 24: (128, 64, 256, 4), 25: (128, 128, 256, 4)}      # (threads = those that issue the DMA requests)


def _halo_fits(tile, H, W):
    """Host restatement of the halo kernels' geometry limits (csrc/gemm.hip: halo_patch / gemm_tile_valid)."""
    bm, bn, threads, nstw = _HALO_GEO[tile]
    tw = 16 if W % 16 == 0 else W
    if tw <= 0 or bm % tw:
        return False
    th = min(bm // tw, H)
    while th > 1 and (H % th or bm % (th * tw)):      # tallest patch that tiles both the image and the block
        th -= 1
    if th <= 0 or H % th or bm % (th * tw):
        return False
    rpp = threads // 8
    hr = (bm // (th * tw)) * (th + 2) * (tw + 2)
    apass, wpass = -(-hr // rpp), -(-bn // rpp)
    return apass <= 12 and (2 * apass * rpp + nstw * wpass * rpp) * 128 + max(bm, hr) * 4 <= 160 * 1024


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K,splitk", [(256, 320, 640, 1), (100, 70, 128, 1), (128, 1280, 2560, 4), (32, 96, 512, 2), (512, 640, 1920, 3),
                                          (300, 192, 64, 1), (512, 1280, 11520, 6)])
def test_gemm(tile, M, N, K, splitk):
    E = _eng()
    a = bf(rnd((M, K), 1)).cuda()
    w = bf(rnd((N, K), 2)).cuda()
    c = torch.full((M, N), float("nan"), device="cuda")
    rc = E.lib(PREC).df_test_gemm(ptr(a), ptr(w), ptr(c), M, N, K, tile, splitk, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t()
    assert torch.isfinite(c).all()
    assert rel_l2(c.cpu(), ref.cpu()) < 2e-3


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K1,K2,splitk", [(200, 320, 1280, 320, 1), (72, 192, 512, 128, 2), (512, 1280, 5120, 1280, 4),
                                              (64, 64, 64, 64, 1)])
def test_gemm_two_operand_tensors(tile, M, N, K1, K2, splitk):
    """K columns [0, K1) from A, [K1, K1+K2) from A2: the merged FF2 + proj_out GEMM (engine.hip st.ffproj)."""
    E = _eng()
    a = bf(rnd((M, K1), 21)).cuda()
    a2 = bf(rnd((M, K2), 22)).cuda()
    w = bf(rnd((N, K1 + K2), 23) / (K1 + K2) ** 0.5).cuda()
    c = torch.full((M, N), float("nan"), device="cuda")
    rc = E.lib(PREC).df_test_gemm_dual(ptr(a), ptr(a2), ptr(w), ptr(c), M, N, K1, K2, tile, splitk, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    ref = torch.cat([a, a2], 1).float() @ w.float().t()
    assert torch.isfinite(c).all()
    assert rel_l2(c.cpu(), ref.cpu()) < 2e-3


@pytest.mark.parametrize("tile", [0, 3, 4, 8, 13, 18, 19, 20, 26, 27, 28, 29])
@pytest.mark.parametrize("splitk", [1, 4])
@pytest.mark.parametrize("act,res,out_operand", [(0, 0, 0), (1, 0, 0), (1, 0, 1), (2, 0, 0), (0, 1, 0), (1, 1, 1), (2, 1, 0)])
def test_gemm_epilogue_times_splitk(tile, splitk, act, res, out_operand):
    """Every simple epilogue feature (bias, residual, SiLU / ReLU, operand-type output) must give the same answer from the
    in-kernel epilogue and from the split-K reduce kernel (round 2: SiLU was missing from the reduce, only the autotuned
    time-embedding GEMM ever took that path)."""
    E = _eng()
    M, N, K = 72, 320, 1280
    a = bf(rnd((M, K), 11)).cuda()
    w = bf(rnd((N, K), 12) / K ** 0.5).cuda()
    bias = rnd((N,), 13).cuda()
    r = rnd((M, N), 14).cuda() if res else None
    c = torch.full((M, N), float("nan"), device="cuda", dtype=odt() if out_operand else torch.float32)
    rc = E.lib(PREC).df_test_gemm_epi(ptr(a), ptr(w), ptr(bias), ptr(r) if res else None, ptr(c), M, N, K, act, out_operand,
                                      tile, splitk, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t() + bias
    if res:
        ref = ref + r
    ref = F.silu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    assert torch.isfinite(c.float()).all()
    assert rel_l2(c.float().cpu(), ref.cpu()) < (6e-3 if out_operand else 2e-3)


@pytest.mark.parametrize("tile", [0, 1, 3, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16, 17, 18, 19, 20, 23, 24, 25, 26, 27, 28, 29])
@pytest.mark.parametrize("NB,H,W,Cin,Cout,stride,ups,splitk", [
    (2, 16, 64, 128, 96, 1, 0, 2), (8, 2, 8, 256, 64, 1, 0, 4), (3, 4, 16, 64, 192, 1, 0, 1), (1, 16, 16, 64, 64, 1, 0, 1),
    (2, 16, 64, 64, 64, 1, 0, 1), (2, 8, 32, 128, 192, 1, 0, 1), (1, 16, 64, 64, 64, 2, 0, 1),
    (2, 4, 16, 128, 64, 1, 1, 1), (2, 2, 8, 256, 320, 1, 0, 4), (3, 8, 16, 64, 4, 1, 0, 1),
    # halo-pass counts 9 and 8 of the producer-specialised tiles (the spread halo schedule counts its waits per pass: the shapes
    # above give 6, 7, 10 and 11)
    (4, 2, 16, 128, 64, 1, 0, 1), (4, 4, 8, 128, 64, 1, 0, 2)])
def test_conv3x3(tile, NB, H, W, Cin, Cout, stride, ups, splitk):
    E = _eng()
    if tile in HALO + (18, 19, 20, 26, 27, 28, 29) and (stride != 1 or ups):
        pytest.skip("halo and producer-specialised kernels cover stride-1 convs only")
    x = bf(rnd((NB, Cin, H, W), 3))
    w = bf(rnd((Cout, Cin, 3, 3), 4) / (3 * Cin ** 0.5))
    b = rnd((Cout,), 5)
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), b, stride=stride, padding=1)
    a = x.permute(0, 2, 3, 1).contiguous().cuda()                       # NHWC bf16
    wp = w.permute(0, 2, 3, 1).contiguous().cuda()                      # [O][ky][kx][I]
    OH, OW = ref.shape[2], ref.shape[3]
    c = torch.full((NB * OH * OW, Cout), float("nan"), device="cuda")
    bc = b.cuda()
    rc = E.lib(PREC).df_test_conv3x3(ptr(a), ptr(wp), ptr(bc), ptr(c), NB, H, W, Cin, Cout, stride, ups, tile, splitk,
                                 stream())
    if rc != 0 and tile in HALO and b"invalid argument" in E.lib(PREC).df_last_error():
        # only shapes whose (th+2)x(tw+2) halo genuinely exceeds the 12 DMA passes / 160 KB LDS may be refused
        assert not _halo_fits(tile, H, W), f"halo tile {tile} refused a {H}x{W} map that fits"
        pytest.skip("patch geometry of this halo tile does not fit LDS for this shape")
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    got = c.cpu().reshape(NB, OH, OW, Cout).permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < 2e-3


@pytest.mark.parametrize("NB,H,W,Cin,Cout", [(2, 16, 64, 128, 3), (1, 5, 36, 64, 1), (3, 2, 8, 256, 4), (1, 128, 512, 128, 3), (2, 3, 100, 32, 2)])
def test_conv3x3_few_output_channels(NB, H, W, Cin, Cout):
    """The VAE decoder's conv_out (Decoder.conv_out: 128 -> 3 channels) as its own kernel: NHWC operand-type input, NCHW fp32 output,
    ragged row tiles (W not a multiple of 32), image borders."""
    E = _eng()
    x = bf(rnd((NB, Cin, H, W), 13))
    w = bf(rnd((Cout, Cin, 3, 3), 14) / (3 * Cin ** 0.5))
    b = rnd((Cout,), 15)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    a = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = w.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.full((NB, Cout, H, W), float("nan"), device="cuda")
    rc = E.lib(PREC).df_test_conv3x3_fewout(ptr(a), ptr(wp), ptr(b.cuda()), ptr(out), NB, H, W, Cin, Cout, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert rel_l2(out.cpu(), ref) < 2e-3


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 17, 18, 19, 20, 23, 24, 25, 26, 27, 28, 29])
@pytest.mark.parametrize("NB,H,W,Cin,Cin2,Cout,splitk", [
    (2, 16, 64, 128, 192, 96, 1), (2, 16, 64, 64, 128, 320, 2), (4, 8, 32, 128, 64, 128, 1), (8, 4, 16, 128, 320, 192, 2),
    (8, 2, 8, 256, 128, 64, 4), (3, 4, 16, 64, 64, 64, 1),
    # long one-tap tails of the producer-specialised halo tiles (23-25): 10 / 15 skip slices, three-slot ring wraps several times
    (2, 16, 64, 128, 640, 128, 1), (8, 16, 64, 320, 960, 320, 1), (8, 4, 16, 1280, 2560, 256, 6)])
def test_conv3x3_with_folded_skip(tile, NB, H, W, Cin, Cin2, Cout, splitk):
    """ResBlock conv2 with a channel change: conv3x3(h) + conv1x1(x) + bias as ONE implicit GEMM over K = 9 Cin + Cin2
    (openai_unetmodel.py:255-275: skip_connection(x) + h): the 1x1 is a tenth K range of the generic stride-1 kernel."""
    E = _eng()
    h = bf(rnd((NB, Cin, H, W), 31))
    x = bf(rnd((NB, Cin2, H, W), 32))
    w3 = bf(rnd((Cout, Cin, 3, 3), 33) / (3 * Cin ** 0.5))
    w1 = bf(rnd((Cout, Cin2, 1, 1), 34) / Cin2 ** 0.5)
    b = rnd((Cout,), 35)
    ref = F.conv2d(h.float(), w3.float(), b, padding=1) + F.conv2d(x.float(), w1.float())
    a = h.permute(0, 2, 3, 1).contiguous().cuda()
    a2 = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = torch.cat([w3.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin), w1.reshape(Cout, Cin2)], 1).contiguous().cuda()
    c = torch.full((NB * H * W, Cout), float("nan"), device="cuda")
    bc = b.cuda()
    rc = E.lib(PREC).df_test_conv3x3_skip(ptr(a), ptr(a2), ptr(wp), ptr(bc), ptr(c), NB, H, W, Cin, Cin2, Cout, tile, splitk, stream())
    if rc != 0 and b"refused" in E.lib(PREC).df_last_error():
        pytest.skip("tile / split-K combination does not exist for this shape (the halo tiles do not take the folded skip)")
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    got = c.cpu().reshape(NB, H, W, Cout).permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < 2e-3


@pytest.mark.parametrize("N,HW,C,silu,eps", [(2, 1024, 320, 1, 1e-5), (3, 128, 64, 0, 1e-6), (1, 512, 960, 1, 1e-5),
                                             (1, 65536, 128, 1, 1e-6),
                                             # register kernel: 1 / 3 / 15 / 40 channel pairs per group, row counts that do not
                                             # divide into its passes, the 20-pass variant, a 1-row tensor
                                             (2, 1000, 64, 1, 1e-5), (1, 777, 192, 0, 1e-5), (2, 1090, 960, 1, 1e-5), (3, 16, 2560, 1, 1e-5),
                                             (2, 1, 320, 0, 1e-5), (1, 5461, 192, 1, 1e-6),
                                             # chunked three-launch form: 256 / 512 channels, a ragged last chunk
                                             (1, 16384 + 100, 256, 1, 1e-6), (2, 8192 + 33, 512, 0, 1e-6)])
def test_groupnorm(N, HW, C, silu, eps):
    E = _eng()
    x = rnd((N, C, HW), 6) * 2 + 0.5
    g, b = rnd((C,), 7), rnd((C,), 8)
    ref = F.group_norm(x, 32, g, b, eps)
    if silu:
        ref = F.silu(ref)
    xin = x.permute(0, 2, 1).contiguous().cuda()
    out = torch.empty(N, HW, C, dtype=odt(), device="cuda")
    gc, bc = g.cuda(), b.cuda()
    rc = E.lib(PREC).df_test_groupnorm(ptr(xin), C, N, HW, C, ptr(gc), ptr(bc), eps, silu, ptr(out), stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu().permute(0, 2, 1), ref) < 4e-3     # bf16 output rounding


@pytest.mark.parametrize("N,HW,C,c_own,nslab,silu", [(2, 1024, 320, 320, 2, 1), (2, 16, 1280, 1280, 8, 1), (1, 256, 960, 640, 4, 1),
                                                     (3, 64, 2560, 1280, 32, 0), (2, 1024, 640, 320, 3, 1)])
def test_groupnorm_finishes_its_producers_split_k(N, HW, C, c_own, nslab, silu):
    """The norm that follows a split-K GEMM does that GEMM's reduce (csrc/elementwise.hip GnSlabs::own): channels [0, c_own) of
    x = sum of the slabs (slab order) + bias + residual, BIT-equal to the same fp32 additions done one after the other, written
    back to x; channels [c_own, C) (the skip half of a concat buffer, 960 = 640 + 320 puts group 21 across the boundary) come
    from x; the normalised operand matches torch's group_norm of the finished tensor."""
    E = _eng()
    rows = N * HW
    slabs = rnd((nslab, rows, c_own), 61) * 0.7
    bias, res = rnd((c_own,), 62), rnd((rows, c_own + 8), 63)
    x0 = rnd((rows, C), 64) * 2 + 0.5
    g, b = rnd((C,), 65), rnd((C,), 66)
    fin = slabs[0].clone()
    for s_ in range(1, nslab):
        fin += slabs[s_]
    fin = fin + bias
    fin = fin + res[:, :c_own]
    full = x0.clone()
    full[:, :c_own] = fin
    ref = F.group_norm(full.reshape(N, HW, C).permute(0, 2, 1), 32, g, b, 1e-5)
    if silu:
        ref = F.silu(ref)
    xc, sc, bc, rc_, gc, btc = x0.cuda(), slabs.cuda(), bias.cuda(), res.cuda(), g.cuda(), b.cuda()
    out = torch.empty(N, HW, C, dtype=odt(), device="cuda")
    rc = E.lib(PREC).df_test_groupnorm_own_slabs(ptr(xc), C, N, HW, C, ptr(gc), ptr(btc), 1e-5, silu, ptr(out), ptr(sc), nslab, c_own,
                                                 ptr(bc), ptr(rc_), c_own + 8, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert torch.equal(xc.cpu(), full)
    assert rel_l2(out.float().cpu().permute(0, 2, 1), ref) < 4e-3


# row lengths beyond the Stage-2 models' (320 / 640 / 1280; classifier 128 / 256 / 512): every multiple of 64 up to 2048 -- a UNet with
# model_channels 256 x channel_mult 4 or 192 x 2 reaches the kernel with 1024- / 384-wide rows on its 2- and 3-token maps (round-6 wide
# sweep of tests/test_unet_config_fuzz_gpu.py: "op 93 (layernorm) failed: invalid argument")
@pytest.mark.parametrize("rows,C", [(1024, 320), (77, 640), (16, 1280), (128, 64), (8192, 320), (3, 128), (5, 256),
                                    (7, 512), (513, 1280), (2, 192), (3, 384), (6, 768), (2, 1024), (9, 960), (5, 2048), (4, 1344)])
def test_layernorm(rows, C):
    E = _eng()
    x = rnd((rows, C), 9) * 3 - 1
    g, b = rnd((C,), 10), rnd((C,), 11)
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    xc, gc, bc = x.cuda(), g.cuda(), b.cuda()
    out = torch.empty(rows, C, dtype=odt(), device="cuda")
    rc = E.lib(PREC).df_test_layernorm(ptr(xc), rows, C, ptr(gc), ptr(bc), ptr(out), stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), ref) < 4e-3


@pytest.mark.parametrize("N,heads,D,Tq,Tk", [(2, 8, 40, 1024, 1024), (2, 8, 80, 256, 256), (1, 8, 160, 64, 64),
                                             (2, 8, 160, 16, 16), (2, 8, 40, 1024, 32), (2, 8, 32, 128, 33),
                                             (1, 2, 64, 256, 256), (1, 2, 128, 64, 32), (1, 8, 80, 256, 32),
                                             # head dims of other model_channels / num_heads quotients (round 6, closing session)
                                             (2, 4, 48, 256, 256), (1, 8, 24, 128, 128), (2, 2, 96, 64, 64), (1, 12, 16, 1024, 1024),
                                             (2, 8, 56, 256, 32), (1, 8, 72, 64, 33), (2, 4, 112, 16, 16), (1, 2, 192, 256, 256),
                                             (1, 4, 96, 1024, 1024), (2, 2, 192, 16, 9), (1, 4, 48, 2, 2), (1, 8, 24, 3, 3)])
def test_attention(N, heads, D, Tq, Tk):
    E = _eng()
    C_ = heads * D
    q = bf(rnd((N, Tq, C_), 12))
    k = bf(rnd((N, Tk, C_), 13))
    v = bf(rnd((N, Tk, C_), 14))
    scale = D ** -0.5
    sp = lambda t, T: t.float().reshape(N, T, heads, D).permute(0, 2, 1, 3)
    att = torch.softmax(sp(q, Tq) @ sp(k, Tk).transpose(-1, -2) * scale, dim=-1)
    ref = (att @ sp(v, Tk)).permute(0, 2, 1, 3).reshape(N, Tq, C_)
    ldvt = (Tk + 31) // 32 * 32
    vt = torch.full((N, C_, ldvt), float("nan"), dtype=odt())      # padding deliberately poisoned
    vt[:, :, :Tk] = v.permute(0, 2, 1)
    qc, kc, vc = q.cuda(), k.cuda(), vt.cuda()
    o = torch.empty(N, Tq, C_, dtype=odt(), device="cuda")
    rc = E.lib(PREC).df_test_attention(ptr(qc), C_, ptr(kc), C_, ptr(vc), ldvt, ptr(o), C_, N, heads, D, Tq, Tk, scale,
                                   stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    assert rel_l2(o.float().cpu(), ref) < 1e-2      # P and O are rounded to bf16 inside the kernel


def test_attention_online_softmax_rescale():
    """Force the running-max update: one key far above the rest in a late tile (guide rule 26)."""
    E = _eng()
    N, heads, D, T = 1, 1, 64, 256
    q = bf(rnd((N, T, D), 20))
    k = bf(rnd((N, T, D), 21) * 0.1)
    k[0, 200] = q[0, 5] * 4            # spike for query 5 in the 7th key tile
    v = bf(rnd((N, T, D), 22))
    scale = D ** -0.5
    att = torch.softmax(q.float() @ k.float().transpose(-1, -2) * scale, dim=-1)
    ref = att @ v.float()
    vt = v.permute(0, 2, 1).contiguous()
    qc, kc, vc = q.cuda(), k.cuda(), vt.cuda()
    o = torch.empty(N, T, D, dtype=odt(), device="cuda")
    rc = E.lib(PREC).df_test_attention(ptr(qc), D, ptr(kc), D, ptr(vc), T, ptr(o), D, N, heads, D, T, T, scale, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert rel_l2(o.float().cpu(), ref) < 1e-2


def test_sampler_arithmetic():
    E = _eng()
    x, e, e2 = rnd((4, 4, 16, 64), 30).cuda(), rnd((4, 4, 16, 64), 31).cuda(), rnd((8, 4, 16, 64), 32).cuda()
    c = E.cfg_combine(e2, 4.5)
    assert torch.allclose(c, e2[:4] + 4.5 * (e2[4:] - e2[:4]), atol=1e-6)
    l = E.lincomb([(0.3, x), (-1.7, e), (2.0, c)])
    assert torch.allclose(l, 0.3 * x - 1.7 * e + 2.0 * c, atol=1e-5)
    a_t, a_prev, s1m = 0.5, 0.7, (1 - 0.5) ** 0.5
    xp, p0 = E.ddim_update(x, e, a_t, a_prev, 0.0, s1m)
    ref0 = (x - s1m * e) / a_t ** 0.5
    assert torch.allclose(p0, ref0, atol=1e-5)
    assert torch.allclose(xp, a_prev ** 0.5 * ref0 + (1 - a_prev) ** 0.5 * e, atol=1e-5)


def _ln_chain_ref(A0, W0, b0, res, gamma, beta, W1, b1, mode):
    """fp64 reference of the producer/consumer pair with the kernel's operand roundings emulated (tight), plus the
    plain LayerNorm -> Linear of the reference model (attention_openai.py:211-215) on the same t0 (loose)."""
    t0 = (A0.double() @ W0.double().t() + b0.double() + res.double()).float()
    xb = bf(t0).double()
    wg = bf(gamma[None, :] * W1).double()
    mean = t0.double().mean(-1, keepdim=True)
    var = (t0.double() ** 2).mean(-1, keepdim=True) - mean ** 2
    rstd = (var + 1e-5).rsqrt()
    bb = (W1.double() @ beta.double()) + (b1.double() if b1 is not None else 0)
    y_emu = rstd * (xb @ wg.t() - mean * wg.sum(-1)[None, :]) + bb
    y_true = F.layer_norm(t0.double(), (t0.shape[1],), gamma.double(), beta.double(), 1e-5) @ W1.double().t()
    if b1 is not None:
        y_true = y_true + b1.double()
    if mode == 1:
        f = lambda y: y.chunk(2, -1)[0] * F.gelu(y.chunk(2, -1)[1])
        y_emu, y_true = f(y_emu), f(y_true)
    return t0, y_emu.float(), y_true.float()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("M,C,T,tile0,sk0,tile1,sk1", [
    (1024, 320, 256, 3, 1, 3, 1), (1024, 320, 256, 13, 1, 2, 1), (512, 640, 64, 1, 1, 12, 1), (128, 1280, 16, 3, 4, 3, 1),
    (256, 128, 64, 0, 1, 0, 1), (512, 64, 128, 4, 1, 13, 1), (2048, 320, 1024, 9, 1, 8, 1), (128, 1280, 16, 3, 4, 3, 2),
    (512, 1280, 64, 1, 1, 9, 1), (512, 1280, 64, 0, 2, 8, 1), (256, 640, 64, 9, 1, 1, 1),    # 8-wave tiles at C = 1280 (20 slots per row)
    # 21 / 22: the persistent GEGLU kernel (ffn.hip); valid for mode 1 only -- several tiles per block, ragged last row tile
    (8192, 320, 1024, 3, 1, 21, 1), (2048, 640, 256, 3, 1, 21, 1), (512, 1280, 64, 3, 1, 22, 1), (1024, 320, 256, 13, 1, 22, 1),
    (192, 128, 64, 0, 1, 21, 1), (4096, 640, 256, 3, 1, 22, 1),
    # 30 / 31: the same kernel with 8 wavefronts per block (10 / 20 row-statistics slots)
    (8192, 320, 1024, 3, 1, 30, 1), (2048, 640, 256, 3, 1, 30, 1), (512, 1280, 64, 3, 1, 31, 1), (192, 128, 64, 0, 1, 30, 1),
    (1024, 320, 256, 13, 1, 31, 1),
    # 18-20: producer-specialised blocks as producer (PROD epilogue) and as LayerNorm-folded consumer (LNC / GEGLU / V^T epilogues)
    (1024, 320, 256, 18, 1, 19, 1), (512, 1280, 64, 19, 1, 18, 1), (2048, 640, 256, 20, 2, 20, 1), (256, 640, 64, 19, 1, 19, 2),
    (1024, 320, 256, 26, 1, 27, 1), (512, 1280, 64, 27, 1, 28, 1), (2048, 640, 256, 28, 2, 29, 1), (256, 640, 64, 29, 1, 26, 2)])
def test_ln_folded_gemm_chain(mode, M, C, T, tile0, sk0, tile1, sk1):
    """LayerNorm never runs as a kernel in the SpatialTransformer: statistics come out of the producer's epilogue and
    the consumer applies them (csrc/gemm.hip epilogue_block / splitk_reduce_vec_kernel)."""
    E = _eng()
    if sk1 > 1 and mode != 0:
        pytest.skip("split-K consumers exist only for the plain LN-folded projection")
    N1 = {0: C, 1: 8 * C if C <= 320 else 2 * C, 2: 3 * C}[mode]
    bn = {0: 128, 1: 64, 2: 128, 3: 64, 4: 128, 8: 256, 9: 128, 10: 128, 11: 64, 12: 128, 13: 64, 14: 128, 18: 128, 19: 128, 20: 128, 26: 64, 27: 64, 28: 64, 29: 128,
          21: 128, 22: 128, 30: 128, 31: 128}
    if tile1 in (21, 22, 30, 31) and mode != 1:
        pytest.skip("the persistent kernel is the GEGLU projection only")
    if mode == 2 and (2 * C) % bn[tile1] != 0:
        tile1 = 3                                   # the transposed-V columns must start on a tile boundary
    A0 = bf(rnd((M, C), 40))
    W0 = bf(rnd((C, C), 41) / C ** 0.5)
    b0 = rnd((C,), 42) * 0.1
    res = rnd((M, C), 43) + 0.7                     # non-zero row means: the mean term of the fold must cancel them
    gamma, beta = 1 + 0.2 * rnd((C,), 44), 0.2 * rnd((C,), 45)
    W1 = rnd((N1, C), 46) / C ** 0.5
    b1 = None if mode == 2 else 0.1 * rnd((N1,), 47)
    t0_ref, y_emu, y_true = _ln_chain_ref(A0, W0, b0, res, gamma, beta, W1, b1, mode)
    dev = lambda t: None if t is None else t.cuda()
    A0c, W0c, b0c, resc, gc, bc, W1c, b1c = map(dev, (A0, W0, b0, res, gamma, beta, W1, b1))
    t0 = torch.full((M, C), float("nan"), device="cuda")
    ldvt = (T + 31) // 32 * 32
    if mode == 0:
        y = torch.full((M, N1), float("nan"), device="cuda")
    else:
        y = torch.full((M, N1 // 2 if mode == 1 else 2 * C), float("nan"), dtype=odt(), device="cuda")
    vt = torch.zeros(M // T, C, ldvt, dtype=odt(), device="cuda")
    rc = E.lib(PREC).df_test_ln_chain(ptr(A0c), ptr(W0c), ptr(b0c), ptr(resc), ptr(gc), ptr(bc), ptr(W1c),
                                      ptr(b1c) if b1c is not None else None, ptr(t0), ptr(y), ptr(vt), M, C, N1, mode, T,
                                      ldvt, tile0, sk0, tile1, sk1, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert rel_l2(t0.cpu(), t0_ref) < 2e-3
    got = y.float().cpu()
    loose = 2.5e-2 if PREC == "bf16" else 4e-3
    if mode == 2:
        assert torch.isfinite(got).all()
        assert rel_l2(got, y_emu[:, :2 * C]) < 6e-3 and rel_l2(got, y_true[:, :2 * C]) < loose
        v_emu = y_emu[:, 2 * C:].reshape(M // T, T, C).permute(0, 2, 1)
        gv = vt.float().cpu()[:, :, :T]
        assert rel_l2(gv, v_emu) < 6e-3
    else:
        assert torch.isfinite(got).all()
        assert rel_l2(got, y_emu) < (2e-3 if mode == 0 else 6e-3)       # modes 1/2 round the output to the operand type
        assert rel_l2(got, y_true) < loose


@pytest.mark.parametrize("tile1", [10, 12, 11, 13, 0, 3])
def test_fused_qkv_is_byte_reproducible_with_two_blocks_per_cu(tile1):
    """The fused QKV projection at the 320-channel level of the B = 4 CFG step (8192 x 960 x 320, V third stored transposed) on the
    double-buffered tiles -- two or more blocks per CU, i.e. another block's MFMAs next to this block's epilogue.  Round 6: tiles
    10 / 12 lost the bias of single V^T elements at random (packed add with the src1 operand select, csrc/common.h pk_add_hi):
    every repetition must give the bytes of the first one, and those of a one-block-per-CU tile."""
    E = _eng()
    M, C, T = 8192, 320, 1024
    N1 = 3 * C
    A0, W0 = bf(rnd((M, C), 40)).to(odt()).cuda(), bf(rnd((C, C), 41) / C ** 0.5).to(odt()).cuda()
    b0, res = (0.1 * rnd((C,), 42)).cuda(), rnd((M, C), 43).cuda()
    gamma, beta = (1 + 0.2 * rnd((C,), 44)).cuda(), (0.2 * rnd((C,), 45)).cuda()
    W1 = (rnd((N1, C), 46) / C ** 0.5).cuda()

    def run(tile):
        t0 = torch.full((M, C), float("nan"), device="cuda")
        y = torch.full((M, 2 * C), float("nan"), dtype=odt(), device="cuda")
        vt = torch.zeros(M // T, C, T, dtype=odt(), device="cuda")
        rc = E.lib(PREC).df_test_ln_chain(ptr(A0), ptr(W0), ptr(b0), ptr(res), ptr(gamma), ptr(beta), ptr(W1), None, ptr(t0), ptr(y),
                                          ptr(vt), M, C, N1, 2, T, T, 3, 1, tile, 1, stream())
        assert rc == 0, E.lib(PREC).df_last_error()
        torch.cuda.synchronize()
        return y.view(torch.int16), vt.view(torch.int16)

    y_ref, vt_ref = run(0)                  # 128 x 128 tile with the 4-deep ring: 128 KB of LDS, one block per CU
    for _ in range(10):
        y, vt = run(tile1)
        assert torch.equal(y, y_ref)
        assert torch.equal(vt, vt_ref)


@pytest.mark.parametrize("M,N,K,act", [(8, 1280, 320, 1), (8, 20160, 1280, 0), (16, 1280, 1280, 1), (2, 130, 256, 2),
                                       (3, 7, 64, 0)])
def test_linear_rows_lds(M, N, K, act):
    E = _eng()
    a = rnd((M, K), 50)
    w = bf(rnd((N, K), 51) / K ** 0.5)
    b = rnd((N,), 52)
    ref = a @ w.float().t() + b
    ref = F.silu(ref) if act == 1 else (torch.sigmoid(ref) if act == 2 else ref)
    ac, wc, bc = a.cuda(), w.cuda(), b.cuda()
    out = torch.full((M, N), float("nan"), device="cuda")
    rc = E.lib(PREC).df_test_linear_rows(ptr(ac), K, None, 0, ptr(wc), ptr(bc), ptr(out), N, M, N, K, act, 1, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref) < 1e-5


def test_time_embed_first_layer_fused():
    """[timestep embedding (util.py:151-171) with the CFG batch duplication] -> Linear -> SiLU in one launch."""
    E = _eng()
    B, K, N = 4, 320, 1280
    t = torch.tensor([961.0, 37.5, 1.0, 500.25])
    half = K // 2
    freqs = torch.exp(-torch.log(torch.tensor(10000.0)) * torch.arange(half, dtype=torch.float32) / half)
    ang = t[:, None] * freqs[None]
    emb = torch.cat([torch.cos(ang), torch.sin(ang)], -1).repeat(2, 1)          # rows m -> t[m % B]
    w = bf(rnd((N, K), 53) / K ** 0.5)
    b = rnd((N,), 54)
    ref = F.silu(emb @ w.float().t() + b)
    tc, wc, bc = t.cuda(), w.cuda(), b.cuda()
    out = torch.full((2 * B, N), float("nan"), device="cuda")
    rc = E.lib(PREC).df_test_linear_rows(None, 0, ptr(tc), B, ptr(wc), ptr(bc), ptr(out), N, 2 * B, N, K, 1, 1, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 2e-4       # cos/sin of arguments up to ~1e3 rad: fp32 argument reduction



# ---- nearest-x2 upsample + conv3x3 as four 2x2-tap convs on the input-resolution map (csrc/gemm_m3.hip) ---------------------
def _ups4_weights(w):
    """[a][b][O][dy][dx][I] from OIHW fp32: the 3x3 taps that land on one input pixel are summed (S_0 = ({0},{1,2}), S_1 = ({0,1},{2}))."""
    S = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
    O, I = w.shape[:2]
    out = torch.zeros(2, 2, O, 2, 2, I)
    for a in (0, 1):
        for b in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    out[a, b, :, dy, dx, :] = sum(w[:, :, ky, kx] for ky in S[a][dy] for kx in S[b][dx])
    return out


@pytest.mark.parametrize("tile,splitk", [(3, 1), (13, 1), (12, 1), (11, 1), (10, 1), (8, 1), (9, 1), (3, 2), (12, 4), (8, 2)])
@pytest.mark.parametrize("NB,H,W,Cin,Cout", [(2, 8, 32, 128, 192), (3, 4, 16, 256, 64), (1, 2, 8, 64, 320), (2, 5, 7, 64, 64)])
def test_upsample_conv_phase_decomposed(tile, splitk, NB, H, W, Cin, Cout):
    """Two references: (i) exact -- the same per-phase weights (fp32 sums rounded once to the operand type) applied as 2x2
    convs, fp32 accumulation: only the summation order differs (2e-3); (ii) the reference module's form -- nearest x2 then
    conv3x3 with the UNROUNDED fp32 weights (openai_unetmodel.py:100-119): operand rounding of the summed weights only."""
    E = _eng()
    if splitk > 1 and 4 * Cin // 64 // splitk < 2:
        pytest.skip("K too short for this split")
    x = bf(rnd((NB, Cin, H, W), 3))
    w = rnd((Cout, Cin, 3, 3), 4) / (3 * Cin ** 0.5)
    b = rnd((Cout,), 5)
    w4 = bf(_ups4_weights(w)).float()                                        # what the engine's packing must produce
    xp = F.pad(x.float(), (1, 1, 1, 1))
    ref = torch.empty(NB, Cout, 2 * H, 2 * W)
    for a in (0, 1):
        for bb in (0, 1):
            k = w4[a, bb].permute(0, 3, 1, 2)                                # [O][I][dy][dx]
            ref[:, :, a::2, bb::2] = F.conv2d(xp[:, :, a:a + H + 1, bb:bb + W + 1], k, b)
    ref_module = F.conv2d(F.interpolate(x.float(), scale_factor=2, mode="nearest"), w, b, padding=1)
    a_dev = x.permute(0, 2, 3, 1).contiguous().cuda()
    w_dev, b_dev = w.cuda().contiguous(), b.cuda()
    scratch = torch.empty(16 * Cout * Cin, dtype=odt(), device="cuda")
    c = torch.full((NB * 4 * H * W, Cout), float("nan"), device="cuda")
    rc = E.lib(PREC).df_test_conv3x3_ups4(ptr(a_dev), ptr(w_dev), ptr(b_dev), ptr(c), ptr(scratch), NB, H, W, Cin, Cout, tile,
                                          splitk, stream())
    assert rc == 0, E.lib(PREC).df_last_error()
    torch.cuda.synchronize()
    got = c.cpu().reshape(NB, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) < 2e-3
    assert rel_l2(got, ref_module) < (6e-3 if PREC == "bf16" else 8e-4)
    packed = scratch.float().cpu().reshape(2, 2, Cout, 2, 2, Cin)
    assert torch.equal(packed, w4)                                           # the packing kernel, bit for bit



# ------------------------------------------------------------------------------------------- wide GEGLU tiles (ffn_wide.hip, round 6)
def _geglu_ref(x, W, cs, bias, eps=1e-5):
    """LayerNorm-folded GEGLU projection on the (32 x | 32 gate) packing, fp32: the arithmetic of attention_openai.py:37-64 behind
    norm3 (:215) as the engine folds it -- rstd (A . W - mean cs) + b, then x * gelu(gate) with the exact-erf GELU."""
    M, C = x.shape
    A = bf(x).float()
    mean = x.mean(1, keepdim=True)
    rstd = torch.rsqrt((x * x).mean(1, keepdim=True) - mean * mean + eps)
    v = rstd * (A @ bf(W).float().t() - mean * cs[None]) + bias[None]
    v = v.view(M, -1, 2, 32)                       # [M][group][x | gate][32]
    return (v[:, :, 0] * F.gelu(v[:, :, 1])).reshape(M, -1)


@pytest.mark.parametrize("tile", [32, 33, 34, 30, 22])      # the three wide tiles; 30 / 22: the persistent kernel on the same data
@pytest.mark.parametrize("M,C", [(512, 320), (256, 640), (200, 1280), (64, 320), (1024, 320)])
def test_geglu_projection_wide_tiles(tile, M, C):
    """ffn_wide.hip against fp32: ragged last row tile (M = 200), one row tile (M = 64), 5 / 10 / 20 K steps, several column tiles;
    the 320-column packing is made from the (32 | 32) one by launch_pack_w320 inside the test entry, as the engine does."""
    E = _eng()
    L = E.lib(PREC)
    N1 = 8 * C
    x = rnd((M, C), 11) * 1.5 + 0.3
    W = rnd((N1, C), 12) * 0.06
    cs = bf(W).float().sum(1)                      # column sums of the operand-rounded rows, as the packer computes them
    bias = rnd((N1,), 13) * 0.2
    xs = x.view(M, C // 64, 64)
    stats = torch.stack([xs.sum(2), (xs * xs).sum(2)], dim=-1).contiguous()      # per-row partials over 64-column slots
    ref = _geglu_ref(x, W, cs, bias)
    out = torch.full((M, N1 // 2), float("nan"), device="cuda", dtype=odt())
    xd, wd, sd, cd, bd = bf(x).cuda(), bf(W).cuda(), stats.cuda(), cs.cuda(), bias.cuda()      # (kept alive: the ABI takes raw pointers)
    rc = L.df_test_geglu(ptr(xd), ptr(wd), ptr(sd), ptr(cd), ptr(bd), ptr(out), M, C, N1, tile, 0, stream())
    if rc != 0 and tile in (30,) and C > 640:
        pytest.skip("tile 30 holds 10 row-statistics slots")
    assert rc == 0, L.df_last_error()
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) < (4e-3 if PREC == "bf16" else 1.5e-3)
    # a second launch with other weights at the same addresses re-packs (the test entry permutes per call)
    W2 = rnd((N1, C), 14) * 0.06
    cs2 = bf(W2).float().sum(1)
    wd.copy_(bf(W2))
    cd.copy_(cs2)
    rc = L.df_test_geglu(ptr(xd), ptr(wd), ptr(sd), ptr(cd), ptr(bd), ptr(out), M, C, N1, tile, 0, stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), _geglu_ref(x, W2, cs2, bias)) < (4e-3 if PREC == "bf16" else 1.5e-3)
