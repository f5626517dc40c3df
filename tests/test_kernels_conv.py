"""Implicit-GEMM conv forward / dgrad / wgrad (csrc/k_conv.hip) vs torch fp32 (CPU).
`backend` = host-emulated build on CPU (default) or the real gfx950 library through the C ABI (-m gpu)."""
import pytest
import torch
import torch.nn.functional as F

from physicsinformeddiffusionmodels_amd._lib import ConvDesc, ptr, stream_ptr


def rel(a, b):
    a, b = a.detach().cpu(), b.detach().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CASES = [
    # B, H, C0, C1, Cout, K, stride, pad, transposed
    (2, 16, 8, 0, 8, 3, 1, 1, 0),
    (3, 8, 16, 16, 40, 3, 1, 1, 0),      # concat of two sources, Cout not multiple of 32
    (2, 16, 2, 0, 8, 7, 1, 3, 0),        # init conv (Cin=2, scalar staging path)
    (5, 4, 32, 0, 64, 3, 1, 1, 0),       # several images per tile
    (2, 16, 16, 0, 16, 4, 2, 1, 0),      # downsample
    (2, 8, 16, 0, 16, 4, 2, 1, 1),       # upsample (transposed)
    (2, 16, 24, 0, 2, 1, 1, 0, 0),       # final 1x1 conv
    (130, 1, 32, 0, 72, 1, 1, 0, 0),     # Linear as 1x1 conv on 1x1 images, ragged M
    (1, 64, 32, 0, 32, 3, 1, 1, 0),      # full-width 64 tile
]


CONV7_CASES = [   # the init convolution in the split form with (kx, channel) flattened into K (conv7x7_split_kernel): forward only,
    (2, 64, 2, 0, 32, 7, 1, 3, 0),       # the Darcy model's shape: 4 rows of 64 per tile        (dgrad / wgrad stay on the fp32 kernels)
    (3, 16, 2, 0, 64, 7, 1, 3, 0),       # one 16x16 image per tile, two n-tiles, odd batch
    (2, 32, 4, 0, 32, 7, 1, 3, 0),       # self-conditioning: Cin = 4 -> two k-steps per kernel row
    (1, 8, 2, 0, 32, 7, 1, 3, 0),        # 8x8 image: the 7x7 window reaches past both borders everywhere; tile = 256 pixels > image? (falls back)
    (5, 32, 2, 0, 32, 7, 1, 3, 0),       # 8 rows of 32 per tile
]


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", CONV7_CASES)
def test_conv_7x7_split_form(backend, monkeypatch, B, H, C0, C1, Cout, K, stride, pad, transposed):
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)
    monkeypatch.setenv("PIDM_CONV_SPLIT", "0")           # and the fp32 kernel on the same shapes
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


NT4_CASES = [   # 1x1 convs that take the permuted 128-channel tile (forward resp. dgrad) once the occupancy gate is lowered
    (2, 16, 16, 0, 128, 1, 1, 0, 0),
    (3, 8, 128, 128, 16, 1, 1, 0, 0),
    (1, 16, 32, 0, 256, 1, 1, 0, 0),
]


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", NT4_CASES)
def test_conv_1x1_permuted_tile(backend, monkeypatch, B, H, C0, C1, Cout, K, stride, pad, transposed):
    monkeypatch.setenv("PIDM_NT4_MIN_WGS", "1")
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


G1_CASES = [   # 1x1 convolutions as a split-form GEMM (conv1x1_split_kernel): forward takes it for Cout % 64 == 0, the input gradient for Cin % 64 == 0
    (2, 16, 64, 0, 128, 1, 1, 0, 0),     # 512 pixels = 4 tiles, one group of four 32-channel tiles, two stages; dgrad: groups of two
    (1, 16, 128, 0, 768, 1, 1, 0, 0),    # to_qkv of the 16x16 level: six groups of four; dgrad 768 -> 128: 24 stages
    (2, 8, 64, 64, 64, 1, 1, 0, 0),      # concatenated sources (res_conv of the up path), groups of two both ways
    (8, 4, 96, 0, 192, 1, 1, 0, 0),      # three stages, Cout = 6 tiles in groups of two (192 % 128 != 0); dgrad 192 -> 96: fp32 kernel
    (1, 16, 256, 0, 512, 1, 1, 0, 0),    # 2 pixel tiles x 4 groups: fewer items than the (lowered) workgroup count -> half groups on the packing of four
    (256, 1, 64, 0, 128, 1, 1, 0, 0),    # a linear (1x1 images): never this kernel (the FiLM linears are packed in sub-blocks without pieces)
]


@pytest.mark.parametrize("wgs", [None, "3", "64"])
@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", G1_CASES)
def test_conv_1x1_split_gemm(backend, monkeypatch, wgs, B, H, C0, C1, Cout, K, stride, pad, transposed):
    """wgs = 3: three persistent workgroups walk all items (several pixel tiles and n-groups per workgroup, accumulator restarts);
    and the fp32 kernel on the same shapes."""
    if wgs:
        monkeypatch.setenv("PIDM_STREAM_WGS", wgs)
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)
    if not wgs:
        monkeypatch.setenv("PIDM_CONV_SPLIT", "0")
        test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


STREAM_CASES = [   # 1x1 weight gradients on the LDS-free streaming kernels: ragged pixel counts and channel groups
    (75, 1, 32, 0, 200, 1, 1, 0, 0),     # 128-wide dY operand, last group ragged (72 of 128), 75 pixels (1x1 images)
    (49, 1, 32, 100, 24, 1, 1, 0, 0),    # 128-wide X operand across a concat boundary (32 + 100 channels), Cout < 32, 49 pixels
    (9, 1, 16, 0, 40, 1, 1, 0, 0),       # 32x32 tiles, 9 pixels (fewer than one per wave pair)
    (2, 16, 128, 0, 384, 1, 1, 0, 0),    # both operands wide: dY side taken
    (63, 1, 16, 0, 40, 1, 1, 0, 0),      # last wave gets 15 pixels: the unrolled loop bound must stay wave-uniform (MFMA ignores exec)
]


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", STREAM_CASES)
def test_conv_1x1_stream_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed):
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


STREAM_SPLIT_CASES = STREAM_CASES + [   # the 16-pixel k-steps of the split-form streams (conv_wgrad_1x1_split4_body)
    (17, 1, 128, 0, 32, 1, 1, 0, 0),     # wide X operand, 17 pixels: one full k-step on wave 0 + a one-pixel tail on another wave
    (1, 16, 64, 0, 768, 1, 1, 0, 0),     # to_qkv at 64 channels: six wide dY tiles x two narrow tiles, two 128-pixel tiles
    (3, 8, 256, 256, 128, 1, 1, 0, 0),   # res_conv of the decoder: 512 concatenated input channels (wide X across the boundary)
    (130, 1, 256, 0, 96, 1, 1, 0, 0),    # 130 pixels = 128 + 2: second split holds a two-pixel tile
]


@pytest.mark.parametrize("form", ["split", "fp32"])
@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", STREAM_SPLIT_CASES)
def test_conv_1x1_stream_wgrad_forms(backend, monkeypatch, form, B, H, C0, C1, Cout, K, stride, pad, transposed):
    """1x1 weight gradients with a >= 128-channel operand: the bf16-pipe stream (default) and the fp32-MFMA stream
    (PIDM_WGRAD1X1_SPLIT=0) on the same cases."""
    monkeypatch.setenv("PIDM_WGRAD1X1_SPLIT", "1" if form == "split" else "0")
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


def test_1x1_split_wgrad_xcd_order_is_a_pure_renumbering(backend, monkeypatch):
    """wgrad_xcd_order only renumbers a problem's workgroups (which XCD runs which (split, tile)): with PIDM_WGRAD1X1_XCD=0 every
    workgroup does the same work and the gradients are bit-identical - also for grids that are not multiples of 8."""
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(97)
    for B, H, Cin, Cout in ((3, 16, 128, 96), (5, 8, 64, 384), (1, 16, 256, 256)):
        x = torch.randn(B, Cin, H, H, generator=g)
        dy = torch.randn(B, Cout, H, H, generator=g)
        x0, dyn = nhwc(x).to(dev), nhwc(dy).to(dev)
        d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=1, KW=1, stride=1, pad=0, transposed=0, out_nchw=0, ldo=Cout)
        got = {}
        for xcd in ("1", "0"):
            monkeypatch.setenv("PIDM_WGRAD1X1_XCD", xcd)
            ws = torch.empty(L.pidm_conv_wgrad_ws(d), dtype=torch.uint8, device=dev)
            dw = torch.full((Cout, Cin, 1, 1), float("nan"), device=dev)
            db = torch.full((Cout,), float("nan"), device=dev)
            L.check(L.pidm_conv_wgrad(d, ptr(x0), None, ptr(dyn), Cout, ptr(dw), ptr(db), ptr(ws), st))
            got[xcd] = (dw.cpu(), db.cpu())
        assert torch.equal(got["1"][0], got["0"][0]) and torch.equal(got["1"][1], got["0"][1])
        ref = torch.einsum("bmhw,bnhw->mn", dy.double(), x.double())
        assert float((got["1"][0].double().view(Cout, Cin) - ref).abs().max() / ref.abs().max()) < 2e-6


def test_1x1_split_wgrad_is_as_accurate_as_the_fp32_mfma(backend, monkeypatch):
    """conv_wgrad_1x1_split4_body against a float64 product next to the fp32-MFMA stream on the same data, at the longest pixel
    contraction a split of the Darcy model sees (K = 2048 pixels of the 16 x 16 level's to_qkv with 8 images; both operand roles):
    weight- and bias-gradient errors within 1.25x of the fp32 stream's and below 2e-6 of the result's scale."""
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(99)
    for Cin, Cout in ((128, 384), (256, 64)):
        B, H = 8, 16
        x = torch.randn(B, Cin, H, H, generator=g)
        dy = torch.randn(B, Cout, H, H, generator=g)
        ref = torch.einsum("bmhw,bnhw->mn", dy.double(), x.double())
        refb = dy.double().sum((0, 2, 3))
        x0, dyn = nhwc(x).to(dev), nhwc(dy).to(dev)
        d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=1, KW=1, stride=1, pad=0, transposed=0, out_nchw=0, ldo=Cout)
        errs = {}
        for form in ("split", "fp32"):
            monkeypatch.setenv("PIDM_WGRAD1X1_SPLIT", "1" if form == "split" else "0")
            ws = torch.empty(L.pidm_conv_wgrad_ws(d), dtype=torch.uint8, device=dev)
            dw = torch.full((Cout, Cin, 1, 1), float("nan"), device=dev)
            db = torch.full((Cout,), float("nan"), device=dev)
            L.check(L.pidm_conv_wgrad(d, ptr(x0), None, ptr(dyn), Cout, ptr(dw), ptr(db), ptr(ws), st))
            errs[form] = [float((dw.double().cpu().view(Cout, Cin) - ref).abs().max() / ref.abs().max()),
                          float((db.double().cpu() - refb).abs().max() / refb.abs().max())]
        print(f"1x1 weight gradient {Cin} -> {Cout}, 2048 pixels: max error / max |reference| (dW, dbias):", errs)
        for e_split, e_fp32, what in zip(errs["split"], errs["fp32"], ("dW", "dbias")):
            assert e_split < 1.25 * e_fp32 + 1e-8, (what, errs)
            assert e_split < 2e-6, (what, errs)


PERSIST_CASES = [   # 3x3 convs walked persistently (several m-tiles per workgroup) once the slot count is lowered
    (6, 16, 32, 0, 32, 3, 1, 1, 0),      # 12 m-tiles -> 3 workgroups x 4 tiles, two Cin chunks
    (5, 8, 16, 16, 64, 3, 1, 1, 0),      # two images per tile, ragged last workgroup, two n-tiles, concat source
    (1, 64, 16, 0, 8, 3, 1, 1, 0),       # wide rows (Wv = 64), single chunk
]


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", PERSIST_CASES)
def test_conv_3x3_persistent(backend, monkeypatch, B, H, C0, C1, Cout, K, stride, pad, transposed):
    monkeypatch.setenv("PIDM_PERSIST_SLOTS", "3")
    monkeypatch.setenv("PIDM_CONV_STREAM", "0")          # the streaming / split kernels would take the eligible shapes
    monkeypatch.setenv("PIDM_CONV_SPLIT", "0")
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


STREAM3_CASES = [   # the streaming persistent 3x3 kernel (two LDS buffers, one barrier per (tile, chunk) stage)
    (6, 16, 32, 32, 64, 3, 1, 1, 0),     # concat source = 2 chunks, 2 n-tiles, 24 work items over 5 workgroups (ragged last one)
    (3, 8, 64, 0, 32, 3, 1, 1, 0),       # two 8x8 images per tile, ragged batch (3 images), single work item per workgroup
    (1, 64, 32, 0, 32, 3, 1, 1, 0),      # full-width 64 tile, single chunk: cross-tile pipeline only
    (2, 32, 96, 0, 32, 3, 1, 1, 0),      # three chunks per tile
    (1, 4, 32, 0, 32, 3, 1, 1, 0),       # one 4x4 image: 16 valid pixels in the tile, everything else padding
]


@pytest.mark.parametrize("wgs", ["5", "256"])
@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", STREAM3_CASES)
def test_conv_3x3_streaming(backend, monkeypatch, wgs, B, H, C0, C1, Cout, K, stride, pad, transposed):
    """The fp32-MFMA streaming kernel and the row-staged fp32 wgrad (the split forms off)."""
    monkeypatch.setenv("PIDM_STREAM_WGS", wgs)
    monkeypatch.setenv("PIDM_CONV_SPLIT", "0")
    monkeypatch.setenv("PIDM_WGRAD_SPLIT", "0")
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


SPLIT_CASES = STREAM3_CASES[:4] + [
    (5, 8, 32, 0, 64, 3, 1, 1, 0),       # 8x8 images, ragged batch, two n-tiles
    (2, 16, 64, 32, 32, 3, 1, 1, 0),     # concat source with unequal parts, 6 chunks of 16 channels
]


@pytest.mark.parametrize("ws", ["1", "0"])
@pytest.mark.parametrize("nw,wgs,p,maxns", [("8", "3", "256", "2"), ("4", "256", "128", "1"), ("4", "2", "256", "3")])
@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", SPLIT_CASES)
def test_conv_3x3_split_forms(backend, monkeypatch, ws, nw, wgs, p, maxns, B, H, C0, C1, Cout, K, stride, pad, transposed):
    """3x3 forward / dgrad / wgrad on the bf16 matrix pipe with 3-piece operands: both tile sizes of each kernel, several work
    items per workgroup (forward) and several tiles per split (wgrad); same tolerance as the fp32-MFMA kernels.  ws = 1: the
    warp-specialised forward / dgrad kernel (producer waves stage, consumer waves compute), ws = 0: every wave does both."""
    monkeypatch.setenv("PIDM_SPLIT_WS", ws)
    monkeypatch.setenv("PIDM_SPLIT_NW", nw)
    monkeypatch.setenv("PIDM_STREAM_WGS", wgs)
    monkeypatch.setenv("PIDM_WGRAD_SPLIT_P", p)
    monkeypatch.setenv("PIDM_WGRAD_SPLIT_MAXNS", maxns)
    # weight gradient: ws = 1 the row-streaming kernel (k_wgrad_rs.hip: no LDS, rolling rows in registers; strips of 2 ... 64 rows,
    # odd strip counts, several strip pairs per wave with maxns), ws = 0 the LDS-staged conv_wgrad_split_kernel it replaced
    monkeypatch.setenv("PIDM_WGRAD_RS", ws)
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


WS_VARIANT_CASES = [   # B, H, C0, C1, Cout, K, stride, pad, transposed (GPU sizes: many workgroups, several stages and items each)
    (16, 16, 128, 0, 128, 3, 1, 1, 0),   # 256-pixel tile by default
    (8, 8, 128, 128, 256, 3, 1, 1, 0),   # 8x8 level: 128-pixel tile, concat source, 16 stages
    (8, 16, 64, 0, 128, 4, 2, 1, 0),     # 4x4 / stride 2 as 4 K-phases
    (8, 8, 128, 0, 64, 4, 2, 1, 1),      # transposed 4x4 / stride 2 as 4 output parities
    (4, 16, 128, 0, 384, 1, 1, 0, 0),    # 1x1 GEMM, four tiles per item
    (4, 8, 256, 0, 192, 1, 1, 0, 0),     # 1x1 GEMM, two tiles per item
]


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", WS_VARIANT_CASES)
def test_split_ws_variants_are_bit_identical(backend, monkeypatch, B, H, C0, C1, Cout, K, stride, pad, transposed):
    """Round 6: the producer waves of the warp-specialised kernels count their vector-memory operations by hand (untracked loads and
    LDS-direct copies, s_waitcnt vmcnt(n) with n > 0) and come in 4 or 8; consumers take one or two sub-tiles.  Same staging, same
    order of the six terms per accumulator: every variant - and every repetition of it, with the previous launches still in flight -
    must reproduce the two-role kernel's output BIT for bit (a copy that had not landed at the stage barrier would show up here)."""
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(77 + H + Cout)
    Cin = C0 + C1
    Ho = H * 2 if transposed else (H + 2 * pad - K) // stride + 1
    x0 = torch.randn(B, H, H, C0, generator=g).to(dev)
    x1 = torch.randn(B, H, H, C1, generator=g).to(dev) if C1 else None
    w = (torch.randn((Cin, Cout, K, K) if transposed else (Cout, Cin, K, K), generator=g) * 0.05).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=K, KW=K, stride=stride, pad=pad, transposed=transposed,
                 out_nchw=0, ldo=Cout)
    wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))

    def run(env, reps):
        for k in ("PIDM_SPLIT_WS", "PIDM_SPLIT_NPW", "PIDM_SPLIT_MS", "PIDM_SPLIT_NW", "PIDM_SPLIT_XCD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs = []
        for _ in range(reps):
            out = torch.full((B, Ho, Ho, Cout), float("nan"), device=dev)
            L.check(L.pidm_conv_forward(d, ptr(x0), ptr(x1), ptr(wp), ptr(bias), None, ptr(out), st))
            outs.append(out)
        return outs
    ref = run({"PIDM_SPLIT_WS": "0"}, 1)[0]
    assert torch.isfinite(ref).all()
    reps = 12 if dev.type == "cuda" else 1
    variants = [{}, {"PIDM_SPLIT_WS": "1", "PIDM_SPLIT_NPW": "4"}, {"PIDM_SPLIT_WS": "1", "PIDM_SPLIT_NPW": "8"},
                {"PIDM_SPLIT_WS": "1", "PIDM_SPLIT_MS": "2"}, {"PIDM_SPLIT_WS": "1", "PIDM_SPLIT_NW": "4", "PIDM_SPLIT_NPW": "8"},
                {"PIDM_SPLIT_XCD": "1"}]
    if K == 1:
        variants = [{}, {"PIDM_SPLIT_NPW": "4"}, {"PIDM_SPLIT_NPW": "8"}]
        if dev.type == "cpu":
            variants = variants[:2]
    elif dev.type == "cpu":
        variants = variants[:1] + variants[3:5]                 # (the emulator runs one fiber per GPU thread: keep the CPU leg short)
    for env in variants:
        for out in run(env, reps):
            assert torch.equal(out, ref), env


RS_CASES = [   # the row-streaming forward / dgrad kernel (k_conv_rs.hip): rows of 32 / 64 pixels, 32 / 64 input channels
    (1, 64, 32, 0, 32, 3, 1, 1, 0),      # two strips per row, one n-tile; dgrad the same
    (3, 32, 32, 32, 64, 3, 1, 1, 0),     # concat source = 4 chunks, two n-groups, odd batch; dgrad: 64 -> 64
    (2, 32, 32, 0, 64, 3, 1, 1, 0),      # two n-tiles per wave; dgrad: 4 chunks -> 32
    (2, 64, 64, 0, 32, 3, 1, 1, 0),      # 4 chunks from one source; dgrad: two n-tiles per wave
]


@pytest.mark.parametrize("waves", ["1", "24", "100000"])
@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", RS_CASES)
def test_conv_3x3_row_streaming(backend, monkeypatch, waves, B, H, C0, C1, Cout, K, stride, pad, transposed):
    """Strips of the whole image height (waves = 1: R = 64 or 32, both values of R % 3), of a few rows, and of 4 rows (every row an
    edge row); ragged last workgroup.  The weight gradient and the GroupNorm-backward sums of the input gradient ride along."""
    monkeypatch.setenv("PIDM_CONV_RS_WAVES", waves)
    monkeypatch.setenv("PIDM_CONV_RS_MINR", "4")
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


def test_conv_3x3_row_streaming_is_taken(backend, monkeypatch, capfd):
    """The launcher picks the row-streaming kernel for the wide levels (and PIDM_CONV_RS=0 turns it off)."""
    L, dev = backend
    st = stream_ptr(dev)
    d = ConvDesc(B=1, Hi=32, Wi=32, C0=32, C1=0, ld0=32, ld1=0, Cout=32, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=32)
    x = torch.randn(1, 32, 32, 32, device=dev)
    w = torch.randn(32, 32, 3, 3, device=dev)
    wp = torch.zeros(L.pidm_conv_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
    out = torch.empty(1, 32, 32, 32, device=dev)
    monkeypatch.setenv("PIDM_CONV_RS_WAVES", "8")
    for off, expect in (("1", True), ("0", False)):
        monkeypatch.setenv("PIDM_CONV_RS", off)
        monkeypatch.setenv("PIDM_TRACE_CONV", "1")
        L.check(L.pidm_conv_forward(d, ptr(x), None, ptr(wp), None, None, ptr(out), st))
        if dev.type == "cuda":
            torch.cuda.synchronize()
        assert ("conv3x3_rs_kernel" in capfd.readouterr().err) == expect


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", STREAM3_CASES[:2])
def test_conv_3x3_streaming_off_matches(backend, monkeypatch, B, H, C0, C1, Cout, K, stride, pad, transposed):
    """PIDM_CONV_STREAM=0 keeps the older tilings reachable (A/B measurements)."""
    monkeypatch.setenv("PIDM_CONV_STREAM", "0")
    monkeypatch.setenv("PIDM_CONV_SPLIT", "0")
    monkeypatch.setenv("PIDM_WGRAD_SPLIT", "0")
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


MT2_CASES = [   # 3x3 convs on the 256-pixel workgroup tile (two m-tiles per wave) once the occupancy gate is lowered
    (2, 64, 16, 0, 32, 3, 1, 1, 0),      # 4 rows of 64 per tile
    (3, 16, 16, 16, 64, 3, 1, 1, 0),     # one whole 16x16 image per tile, concat source, 64-channel tile (NT = 2)
    (5, 8, 32, 0, 32, 3, 1, 1, 0),       # four 8x8 images per tile, ragged last tile (5 images)
    (2, 32, 32, 0, 40, 3, 1, 1, 0),      # 8 rows of 32, Cout not a multiple of 32
]


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", MT2_CASES)
def test_conv_3x3_two_mtiles_per_wave(backend, monkeypatch, B, H, C0, C1, Cout, K, stride, pad, transposed):
    monkeypatch.setenv("PIDM_MT2_MIN_WGS", "1")
    monkeypatch.setenv("PIDM_CONV_STREAM", "0")
    monkeypatch.setenv("PIDM_CONV_SPLIT", "0")
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", CASES)
def test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed):
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(1234 + B + H + C0 + Cout)
    Cin = C0 + C1
    x = torch.randn(B, Cin, H, H, generator=g)
    if transposed:
        w = torch.randn(Cin, Cout, K, K, generator=g) / (Cin * 4) ** 0.5
    else:
        w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    bias = torch.randn(Cout, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    if transposed:
        ref = F.conv_transpose2d(xr, wr, br, stride=stride, padding=pad)
    else:
        ref = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    Ho = ref.shape[-1]
    res = torch.randn(B, Cout, Ho, Ho, generator=g)
    dy = torch.randn(B, Cout, Ho, Ho, generator=g)
    (ref + res).backward(dy)

    x0 = nhwc(x[:, :C0]).to(dev)
    x1 = nhwc(x[:, C0:]).to(dev) if C1 else None
    w, bias = w.to(dev), bias.to(dev)
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=K, KW=K, stride=stride, pad=pad,
                 transposed=transposed, out_nchw=0, ldo=Cout)
    # forward
    wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
    out = torch.full((B, Ho, Ho, Cout), float("nan"), device=dev)
    resn = nhwc(res).to(dev)
    L.check(L.pidm_conv_forward(d, ptr(x0), ptr(x1), ptr(wp), ptr(bias), ptr(resn), ptr(out), st))
    assert rel(out, nhwc((ref + res).detach())) < 2e-6
    # NCHW output variant
    d2 = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=K, KW=K, stride=stride, pad=pad,
                  transposed=transposed, out_nchw=1, ldo=Cout)
    out2 = torch.full((B, Cout, Ho, Ho), float("nan"), device=dev)
    L.check(L.pidm_conv_forward(d2, ptr(x0), ptr(x1), ptr(wp), ptr(bias), None, ptr(out2), st))
    assert rel(out2, ref.detach()) < 2e-6
    # dgrad (+ residual add into dx)
    wd = torch.empty(L.pidm_conv_dgrad_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wd), 1, st))
    dx = torch.full((B, H, H, Cin), float("nan"), device=dev)
    extra = torch.randn(B, H, H, Cin, generator=g).to(dev)
    dyn = nhwc(dy).to(dev)
    L.check(L.pidm_conv_dgrad(d, ptr(dyn), Cout, ptr(wd), ptr(extra), ptr(dx), Cin, st))
    assert rel(dx - extra, nhwc(xr.grad)) < 5e-6
    # wgrad + bias grad
    ws = torch.empty(L.pidm_conv_wgrad_ws(d), dtype=torch.uint8, device=dev)
    dw = torch.full_like(w, float("nan"))
    db = torch.full((Cout,), float("nan"), device=dev)
    L.check(L.pidm_conv_wgrad(d, ptr(x0), ptr(x1), ptr(dyn), Cout, ptr(dw), ptr(db), ptr(ws), st))
    assert rel(dw, wr.grad) < 5e-6
    assert rel(db, br.grad) < 5e-6


GN_EPILOGUE_CASES = [
    # B, H, C0, C1, Cout, groups
    (2, 64, 32, 0, 32, 8),      # 64-wide rows: a wave = half a row; 4 channels per group
    (3, 32, 32, 32, 64, 8),     # concat source, two 32-channel accumulators per wave, 8 channels per group
    (2, 16, 64, 0, 128, 8),     # 16-wide rows: a wave = two rows; 16 channels per group
    (5, 8, 128, 0, 256, 8),     # two 8x8 images per tile (ragged last tile); a whole 32-channel tile per group
    (2, 32, 32, 0, 64, 4),      # 16 channels per group on 32-wide rows: the row-streaming kernel's widest in-row group sum
    (1, 64, 64, 0, 32, 2),      # the same with 64 input channels and 64-wide rows
    (2, 32, 32, 0, 32, 1),      # 32 channels per group: a group spans both 16-lane DPP rows of the row-streaming kernel's half wave
    (2, 32, 32, 0, 64, 2),      # the same with two n-tiles per wave
    (1, 64, 64, 0, 32, 1),      # 64-wide rows, 64 input channels
]


@pytest.mark.parametrize("rs", ["1", "0"])
@pytest.mark.parametrize("B,H,C0,C1,Cout,groups", GN_EPILOGUE_CASES)
def test_conv_epilogue_groupnorm_partials(backend, monkeypatch, rs, B, H, C0, C1, Cout, groups):
    """3x3 convolution whose epilogue also leaves the GroupNorm sums of its output (what launch_gn_stats would compute in a second
    pass): per image and group, the partials must add up to sum / sum of squares of the convolution output."""
    import torch.nn.functional as F
    L, dev = backend
    st = stream_ptr(dev)
    monkeypatch.setenv("PIDM_CONV_RS", rs)               # the row-streaming kernel (one chunk per strip) / the tile kernels
    monkeypatch.setenv("PIDM_CONV_RS_WAVES", "8")
    monkeypatch.setenv("PIDM_CONV_RS_MINR", "4")
    g = torch.Generator().manual_seed(77 + H)
    Cin = C0 + C1
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, bias, padding=1)
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=Cout)
    x0 = nhwc(x[:, :C0]).to(dev)
    x1 = nhwc(x[:, C0:]).to(dev) if C1 else None
    wd = w.to(dev)
    wp = torch.zeros(L.pidm_conv_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(wd), ptr(wp), 0, st))
    out = torch.empty(B, H, H, Cout, device=dev)
    chunks_max = H * H // 32
    part = torch.full((B, chunks_max, groups, 2), float("nan"), dtype=torch.float64, device=dev)
    nch = L.lib.pidm_conv_forward_gn_partials(d, ptr(x0), ptr(x1) if C1 else None, ptr(wp), ptr(bias.to(dev)), ptr(out), groups, ptr(part), st)
    # chunks per image: 32 consecutive pixels each behind the tile kernels, whole strips behind the row-streaming kernel
    assert 0 < nch <= chunks_max and chunks_max % nch == 0, L.lib.pidm_last_error().decode()
    assert rel(out.permute(0, 3, 1, 2), ref) < 5e-6
    cpg = Cout // groups
    rg = ref.double().reshape(B, groups, cpg, H * H)
    s1 = rg.sum(dim=(2, 3))
    s2 = (rg * rg).sum(dim=(2, 3))
    pc = part.cpu().reshape(-1)[:B * nch * groups * 2].reshape(B, nch, groups, 2)
    assert torch.isfinite(pc).all()                      # every (image, chunk, group) slot was written
    assert (pc[..., 0].sum(dim=1) - s1).abs().max().item() < 1e-5 * s2.sqrt().max().item() * (H * H * cpg) ** 0.5
    assert rel(pc[..., 1].sum(dim=1), s2) < 2e-6
    if nch == chunks_max:    # per chunk: 32 consecutive pixels of one image
        ck = rg.reshape(B, groups, cpg, chunks_max, 32).sum(dim=(2, 4)).permute(0, 2, 1)
        assert (pc[..., 0] - ck).abs().max().item() < 1e-4 * ck.abs().max().item()


SPLIT2_CASES = [   # the 4x4 / stride-2 family in the split form: strided conv = 4 K-phases, transposed conv = 4 output parities
    (2, 32, 32, 0, 32, 4, 2, 1, 0),      # downsample 32x32 -> 16x16; its dgrad is the parity form
    (3, 16, 64, 0, 32, 4, 2, 1, 0),      # 16x16 -> 8x8, ragged batch (several images per tile), 4 chunks per phase
    (2, 16, 32, 32, 64, 4, 2, 1, 0),     # concat source, two n-tiles
    (2, 16, 32, 0, 32, 4, 2, 1, 1),      # upsample 16x16 -> 32x32 (parity form); its dgrad is the phased form
    (3, 8, 64, 0, 64, 4, 2, 1, 1),       # 8x8 -> 16x16, ragged batch, two n-tiles
]


@pytest.mark.parametrize("ws", ["1", "0"])
@pytest.mark.parametrize("nw,wgs", [("8", "3"), ("4", "256")])
@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", SPLIT2_CASES)
def test_conv_4x4s2_split_forms(backend, monkeypatch, ws, nw, wgs, B, H, C0, C1, Cout, K, stride, pad, transposed):
    monkeypatch.setenv("PIDM_SPLIT_WS", ws)
    monkeypatch.setenv("PIDM_SPLIT_NW", nw)
    monkeypatch.setenv("PIDM_STREAM_WGS", wgs)
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", SPLIT2_CASES[:1] + SPLIT2_CASES[3:4])
def test_conv_4x4s2_fp32_forms(backend, monkeypatch, B, H, C0, C1, Cout, K, stride, pad, transposed):
    monkeypatch.setenv("PIDM_CONV_SPLIT", "0")
    test_conv_fwd_dgrad_wgrad(backend, B, H, C0, C1, Cout, K, stride, pad, transposed)


@pytest.mark.parametrize("B,H,C0,C1,Cout,K,stride,pad,transposed", [SPLIT_CASES[0], SPLIT_CASES[5], SPLIT2_CASES[0], SPLIT2_CASES[3]])
def test_warp_specialised_kernel_is_bit_identical(backend, monkeypatch, B, H, C0, C1, Cout, K, stride, pad, transposed):
    """Producer / consumer waves change who does the work, not the arithmetic or its order: same bits as the one-role kernel."""
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(321)
    Cin = C0 + C1
    x = torch.randn(B, Cin, H, H, generator=g)
    w = (torch.randn(Cin, Cout, K, K, generator=g) if transposed else torch.randn(Cout, Cin, K, K, generator=g)) * 0.05
    bias = torch.randn(Cout, generator=g).to(dev)
    x0 = nhwc(x[:, :C0]).to(dev)
    x1 = nhwc(x[:, C0:]).to(dev) if C1 else None
    wd_ = w.to(dev)
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=K, KW=K, stride=stride, pad=pad, transposed=transposed,
                 out_nchw=0, ldo=Cout)
    Ho = (H - 1) * stride - 2 * pad + K if transposed else (H + 2 * pad - K) // stride + 1
    outs = {}
    for ws in ("1", "0"):
        monkeypatch.setenv("PIDM_SPLIT_WS", ws)
        wp = torch.zeros(L.pidm_conv_packed_weight_floats(d), device=dev)
        L.check(L.pidm_conv_pack_weights(d, ptr(wd_), ptr(wp), 0, st))
        out = torch.full((B, Ho, Ho, Cout), float("nan"), device=dev)
        L.check(L.pidm_conv_forward(d, ptr(x0), ptr(x1), ptr(wp), ptr(bias), None, ptr(out), st))
        wdg = torch.zeros(L.pidm_conv_dgrad_packed_weight_floats(d), device=dev)
        L.check(L.pidm_conv_pack_weights(d, ptr(wd_), ptr(wdg), 1, st))
        dyn = torch.randn(B, Ho, Ho, Cout, generator=torch.Generator().manual_seed(5)).to(dev)
        dx = torch.full((B, H, H, Cin), float("nan"), device=dev)
        L.check(L.pidm_conv_dgrad(d, ptr(dyn), Cout, ptr(wdg), None, ptr(dx), Cin, st))
        outs[ws] = (out.clone(), dx.clone())
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
    assert torch.isfinite(outs["1"][0]).all() and torch.isfinite(outs["1"][1]).all()


@pytest.mark.parametrize("fwd_kernel", ["tile", "row_streaming"])
def test_split_form_is_as_accurate_as_the_fp32_mfma(backend, monkeypatch, fwd_kernel):
    """The 3-piece / 6-term bf16 form against a float64 convolution, next to the fp32-MFMA kernels on the same data (K = 576):
    forward, input-gradient and weight-gradient errors of the split form stay within 1.25x of the fp32-MFMA kernels' and below
    2e-6 of the result's scale (measured on an MI355X: see profiles/r02_split_conv_notes.txt).  fwd_kernel: forward / input
    gradient on conv3x3_split_kernel (chunk-major accumulation) or on conv3x3_rs_kernel (tap-major; k_conv_rs.hip)."""
    L, dev = backend
    st = stream_ptr(dev)
    monkeypatch.setenv("PIDM_CONV_RS", "1" if fwd_kernel == "row_streaming" else "0")
    monkeypatch.setenv("PIDM_CONV_RS_WAVES", "8")
    monkeypatch.setenv("PIDM_CONV_RS_MINR", "4")
    g = torch.Generator().manual_seed(99)
    B, H, Cin, Cout = 2, 32, 64, 64
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    dy = torch.randn(B, Cout, H, H, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, padding=1)
    ref.backward(dy.double())
    x0, dyn, wd_ = nhwc(x).to(dev), nhwc(dy).to(dev), w.to(dev)
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=Cout)
    errs = {}
    for form in ("split", "fp32"):
        monkeypatch.setenv("PIDM_CONV_SPLIT", "1" if form == "split" else "0")
        monkeypatch.setenv("PIDM_WGRAD_SPLIT", "1" if form == "split" else "0")
        wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev)
        L.check(L.pidm_conv_pack_weights(d, ptr(wd_), ptr(wp), 0, st))
        out = torch.empty(B, H, H, Cout, device=dev)
        L.check(L.pidm_conv_forward(d, ptr(x0), None, ptr(wp), None, None, ptr(out), st))
        wdg = torch.empty(L.pidm_conv_dgrad_packed_weight_floats(d), device=dev)
        L.check(L.pidm_conv_pack_weights(d, ptr(wd_), ptr(wdg), 1, st))
        dx = torch.empty(B, H, H, Cin, device=dev)
        L.check(L.pidm_conv_dgrad(d, ptr(dyn), Cout, ptr(wdg), None, ptr(dx), Cin, st))
        ws = torch.empty(L.pidm_conv_wgrad_ws(d), dtype=torch.uint8, device=dev)
        dw = torch.empty_like(wd_)
        L.check(L.pidm_conv_wgrad(d, ptr(x0), None, ptr(dyn), Cout, ptr(dw), None, ptr(ws), st))
        errs[form] = [float((a.double().cpu() - b).abs().max() / b.abs().max()) for a, b in
                      ((out, nhwc(ref.detach())), (dx, nhwc(xr.grad)), (dw, wr.grad))]
    print("max error / max |reference| (forward, dgrad, wgrad):", errs)
    for e_split, e_fp32, what in zip(errs["split"], errs["fp32"], ("forward", "dgrad", "wgrad")):
        assert e_split < 1.25 * e_fp32 + 1e-8, (what, errs)
        assert e_split < 2e-6, (what, errs)


def test_1x1_split_gemm_is_as_accurate_as_the_fp32_mfma(backend, monkeypatch):
    """conv1x1_split_kernel against a float64 product, next to the fp32-MFMA kernel on the same data, at the longest contraction of
    the Darcy model (K = 768: the input gradient of to_qkv; 48 k-steps x 6 terms in two accumulator chains): forward and
    input-gradient errors within 1.25x of the fp32 kernel's and below 2e-6 of the result's scale."""
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(98)
    B, H, Cin, Cout = 2, 16, 768, 128
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    dy = torch.randn(B, Cout, H, H, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None)
    ref.backward(dy.double())
    x0, dyn, wd_ = nhwc(x).to(dev), nhwc(dy).to(dev), w.to(dev)
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=1, KW=1, stride=1, pad=0, transposed=0, out_nchw=0, ldo=Cout)
    errs = {}
    for form in ("split", "fp32"):
        monkeypatch.setenv("PIDM_CONV_SPLIT", "1" if form == "split" else "0")
        wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev)
        L.check(L.pidm_conv_pack_weights(d, ptr(wd_), ptr(wp), 0, st))
        out = torch.empty(B, H, H, Cout, device=dev)
        L.check(L.pidm_conv_forward(d, ptr(x0), None, ptr(wp), None, None, ptr(out), st))
        wdg = torch.empty(L.pidm_conv_dgrad_packed_weight_floats(d), device=dev)
        L.check(L.pidm_conv_pack_weights(d, ptr(wd_), ptr(wdg), 1, st))
        dx = torch.empty(B, H, H, Cin, device=dev)
        L.check(L.pidm_conv_dgrad(d, ptr(dyn), Cout, ptr(wdg), None, ptr(dx), Cin, st))
        errs[form] = [float((a.double().cpu() - b).abs().max() / b.abs().max()) for a, b in
                      ((out, nhwc(ref.detach())), (dx, nhwc(xr.grad)))]
    print("max error / max |reference| (forward K=768, dgrad K=128):", errs)
    for e_split, e_fp32, what in zip(errs["split"], errs["fp32"], ("forward", "dgrad")):
        assert e_split < 1.25 * e_fp32 + 1e-8, (what, errs)
        assert e_split < 2e-6, (what, errs)


@pytest.mark.parametrize("H", [16, 32])       # 32: rows wide enough for the row-streaming forward / input-gradient kernel
@pytest.mark.parametrize("form", ["split", "fp32"])
def test_conv_extreme_magnitudes(backend, monkeypatch, form, H):
    """Non-finite and extreme inputs through the 3x3 kernels, split form next to the fp32-MFMA form (same expectations for both):
    * +-inf / NaN in the input make exactly the outputs of their 3x3 footprint non-finite (the split form turns inf into NaN -
      x - bf16(x) = inf - inf - where the fp32 kernels keep +-inf; a fp32 value above the bf16 maximum 3.3895e38 rounds its
      first piece to inf and behaves like inf) and leave every other output untouched;
    * 3e38 (below the bf16 maximum) is carried exactly by the three pieces: relative error at the fp32 level;
    * values around 1e-38 and fp32 denormals lose their low pieces (bf16 pieces below 2^-133 vanish) - an ABSOLUTE error far
      below anything representable next to O(1) data, and no NaN.
    Same for the input gradient (specials in dy) and the weight gradient (specials in x)."""
    L, dev = backend
    st = stream_ptr(dev)
    monkeypatch.setenv("PIDM_CONV_SPLIT", "1" if form == "split" else "0")
    monkeypatch.setenv("PIDM_WGRAD_SPLIT", "1" if form == "split" else "0")
    monkeypatch.setenv("PIDM_CONV_RS_WAVES", "8")        # (two images: let conv3x3_rs_kernel take the 32-wide case)
    monkeypatch.setenv("PIDM_CONV_RS_MINR", "4")
    g = torch.Generator().manual_seed(7)
    B, Cin, Cout = 2, 32, 32
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    w = torch.where(w.abs() < 1e-3, torch.full_like(w, 1e-3), w)          # no zero weight: inf * w is never NaN in the reference
    specials = [((0, 3, 2, 2), float("inf")), ((0, 7, 2, 8), float("-inf")), ((0, 1, 2, 13), float("nan")),
                ((0, 5, 8, 2), 3e38), ((1, 9, 3, 3), 1e-38), ((1, 2, 3, 4), 1e-40), ((1, 2, 9, 9), -1e-38)]

    def plant(t):
        t = t.clone()
        for (b, c, y, x_), v in specials:
            t[b, c, y, x_] = v
        t[1, :, 11:16, 0:5] *= 1e-38      # a whole patch at the bottom of the fp32 range
        return t

    def check(out, ref, what):
        out, ref = out.detach().cpu(), ref.detach().cpu()
        bad_ref = ~torch.isfinite(ref)
        assert torch.equal(~torch.isfinite(out), bad_ref), what         # the same set of non-finite outputs
        huge = (ref.abs() > 1e30) & ~bad_ref
        ok = ~bad_ref & ~huge
        scale = ref[ok].abs().max().item()
        assert (out[ok] - ref[ok]).abs().max().item() < 5e-6 * scale, what
        if huge.any():
            assert ((out[huge] - ref[huge]).abs() <= 5e-6 * ref[huge].abs()).all(), what
        return int(bad_ref.sum()), int(huge.sum())

    d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=Cout)
    wd_ = w.to(dev)
    # forward
    x = plant(torch.randn(B, Cin, H, H, generator=g))
    ref = F.conv2d(x, w, None, padding=1)
    wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(wd_), ptr(wp), 0, st))
    out = torch.empty(B, H, H, Cout, device=dev)
    xn = nhwc(x).to(dev)                  # (named: the kernels get raw pointers, the tensors must outlive the calls)
    L.check(L.pidm_conv_forward(d, ptr(xn), None, ptr(wp), None, None, ptr(out), st))
    nbad, nhuge = check(out.permute(0, 3, 1, 2), ref, "forward")
    assert nbad == 3 * 9 * Cout and nhuge == 9 * Cout
    # input gradient
    dy = plant(torch.randn(B, Cout, H, H, generator=g))
    dx_ref = torch.nn.grad.conv2d_input((B, Cin, H, H), w, dy, padding=1)
    wdg = torch.empty(L.pidm_conv_dgrad_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(wd_), ptr(wdg), 1, st))
    dx = torch.empty(B, H, H, Cin, device=dev)
    dyn = nhwc(dy).to(dev)
    L.check(L.pidm_conv_dgrad(d, ptr(dyn), Cout, ptr(wdg), None, ptr(dx), Cin, st))
    check(dx.permute(0, 3, 1, 2), dx_ref, "dgrad")
    # weight gradient: a special value in channel c of x reaches every tap of dW[:, c]
    dy2 = 0.1 * torch.randn(B, Cout, H, H, generator=g)     # |3e38 * dy| stays clear of the fp32 overflow threshold, where the
    dy2 = torch.where(dy2.abs() < 1e-3, torch.full_like(dy2, 1e-3), dy2)    # order of the partial sums decides between inf and 3.4e38
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dy2, padding=1)
    ws = torch.empty(L.pidm_conv_wgrad_ws(d), dtype=torch.uint8, device=dev)
    dw = torch.empty_like(wd_)
    dy2n = nhwc(dy2).to(dev)
    L.check(L.pidm_conv_wgrad(d, ptr(xn), None, ptr(dy2n), Cout, ptr(dw), None, ptr(ws), st))
    check(dw, dw_ref, "wgrad")


def test_per_kernel_profile_table(backend):
    """pidm_prof_kernels_begin / _collect (bench.py's `roofline.top_kernels`): one line per kernel name with launches, total ms, the
    FLOPs the launcher declared and its class; the hooks are off again afterwards."""
    import ctypes as C
    L, dev = backend
    st = stream_ptr(dev)
    B, H, Cin, Cout = 2, 16, 32, 32
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=Cout)
    x = torch.randn(B, H, H, Cin).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3) * 0.1).to(dev)
    wp = torch.zeros(L.pidm_conv_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
    out = torch.empty(B, H, H, Cout, device=dev)
    L.check(L.pidm_prof_kernels_begin(st))
    assert L.pidm_prof_kernels_begin(st) != 0                      # already on
    for _ in range(3):
        L.check(L.pidm_conv_forward(d, ptr(x), None, ptr(wp), None, None, ptr(out), st))
    buf = C.create_string_buffer(4096)
    need = L.pidm_prof_kernels_collect(buf, len(buf))
    assert 0 < need <= len(buf)
    rows = [ln.split("\t") for ln in buf.value.decode().splitlines()]
    assert len(rows) == 1 and rows[0][0].startswith("conv") and int(rows[0][1]) == 3 and float(rows[0][2]) >= 0.0
    assert abs(float(rows[0][3]) - 3 * 2.0 * B * H * H * Cout * Cin * 9) < 100.0 and int(rows[0][4]) in (0, 2)   # (%.6e text)
    assert L.pidm_prof_kernels_collect(buf, len(buf)) < 0            # not on any more
