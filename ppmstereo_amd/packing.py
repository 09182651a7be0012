"""Packs reference-layout conv weights into the LDS image the implicit-GEMM kernel stages verbatim.

Reference layout: Conv2d (Cout, Cin, kh, kw) / Conv3d (Cout, Cin, kt, kh, kw) fp32
(/root/reference/models/core/ppmtereo_update.py, e.g. :254-289).

Packed layout (ppmstereo_amd/csrc/conv_gemm2.hip): bf16 [k-step][M/64][2 planes (hi, lo)][64 couts][32 k];
the input segments are concatenated along K (each zero-padded to a multiple of 32 channels), and inside each [64][32]
tile the four 16-byte chunks of row m are stored at chunk position c ^ ((m >> 2) & 3) (the kernels' bank-conflict-free
read swizzle).  One (k-step, cout block) tile is a contiguous 8 KiB block that reaches LDS by a linear copy.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

BM, BK = 64, 32


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def split_bf16(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def pack_conv2(weight: torch.Tensor, bias: Optional[torch.Tensor], seg_channels: Sequence[int],
               seg_padded: Optional[Sequence[int]] = None, cout_map: Optional[Sequence[int]] = None,
               m_pad: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, dict]:
    """Layout of the second-generation kernel (ppmstereo_amd/csrc/conv_gemm2.hip):
    bf16 [k-step][M/64][2 planes][64][32], k-step = ((kz*kh + ky) * nchunk + chunk) * kw + kx -- the kw taps along x
    of one (dt, dy, 32-channel chunk) are consecutive k-steps (they sweep one LDS activation window) and all cout
    blocks of a k-step are contiguous (one workgroup stages them with a single linear copy).  """
    w = weight.detach().float()
    if w.dim() == 4:
        w = w[:, :, None]
    cout, cin, kt, kh, kw = w.shape
    assert sum(seg_channels) == cin, (seg_channels, cin)
    seg_padded = [_pad_to(c, BK) for c in seg_channels] if seg_padded is None else list(seg_padded)
    assert all(p % BK == 0 and p >= c for p, c in zip(seg_padded, seg_channels))
    cpad = sum(seg_padded)
    nchunk = cpad // BK
    rows = list(range(cout)) if cout_map is None else list(cout_map)
    M = _pad_to(max(rows) + 1, BM) if m_pad is None else m_pad
    assert M % BM == 0 and max(rows) < M
    wk = w.permute(0, 2, 3, 4, 1).reshape(cout, kt * kh, kw, cin)              # [cout][trow][kx][ci]
    full = torch.zeros(M, kt * kh, kw, cpad, dtype=torch.float32, device=w.device)
    ridx = torch.tensor(rows, device=w.device)
    src = dst = 0
    for c, p in zip(seg_channels, seg_padded):
        full[ridx, :, :, dst:dst + c] = wk[:, :, :, src:src + c]
        src += c
        dst += p
    nk = kt * kh * nchunk * kw
    t = full.reshape(M // BM, BM, kt * kh, kw, nchunk, 4, 8).permute(2, 4, 3, 0, 1, 5, 6).contiguous()   # [trow][chunk][kx][mblk][m][c][8]
    t = t.reshape(nk, M // BM, BM, 4, 8)
    m = torch.arange(BM, device=w.device)
    chunk = torch.arange(4, device=w.device)
    pos = chunk[None, :] ^ ((m[:, None] >> 2) & 3)
    sw = torch.empty_like(t)
    sw[:, :, m[:, None].expand(BM, 4), pos] = t
    hi, lo = split_bf16(sw)
    packed = torch.stack([hi, lo], dim=2).contiguous()                    # [ks][mblk][2][64][4][8]
    b = torch.zeros(M, dtype=torch.float32, device=w.device)
    if bias is not None:
        b[ridx] = bias.detach().float()
    meta = dict(M=M, nk=nk, taps=(kt, kh, kw), cpad=cpad, seg_padded=seg_padded, version=2)
    return packed.reshape(-1), b, meta


def pack_conv4(weight: torch.Tensor, bias: Optional[torch.Tensor], seg_channels: Sequence[int],
               seg_padded: Optional[Sequence[int]] = None, cout_map: Optional[Sequence[int]] = None,
               m_pad: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, dict]:
    """Layout of the barrier-free kernel (ppmstereo_amd/csrc/conv_gemm5.hip): the weights in MFMA-fragment order, so that a
    wave loads its A operands straight from global memory into registers with coalesced 1 KiB loads:
        bf16 [k16-step][M/64][frag = 2*mb + plane][lane = 32*h + r][8]
             = W[cout = 64*blk + 32*mb + r][cin = 16*chunk + 8*h + j] of tap (row-step, s), plane 0 = hi, 1 = lo,
        k16-step = ((kz*kh + ky) * nchunk16 + chunk) * kw + kx.
    As for conv_gemm2's swept forms the LAST kernel axis is the swept one: y-swept convs are passed with kh / kw swapped, 2-D swept ones
    with (ky, kx) flattened into x."""
    w = weight.detach().float()
    if w.dim() == 4:
        w = w[:, :, None]
    cout, cin, kt, kh, kw = w.shape
    assert sum(seg_channels) == cin, (seg_channels, cin)
    seg_padded = [_pad_to(c, 16) for c in seg_channels] if seg_padded is None else list(seg_padded)
    assert all(p % 16 == 0 and p >= c for p, c in zip(seg_padded, seg_channels))
    cpad = sum(seg_padded)
    nchunk = cpad // 16
    rows = list(range(cout)) if cout_map is None else list(cout_map)
    M = _pad_to(max(rows) + 1, 128) if m_pad is None else m_pad
    assert M % 64 == 0 and max(rows) < M                    # (the kernel serves 128, 192 and 256 rows)
    wk = w.permute(0, 2, 3, 4, 1).reshape(cout, kt * kh, kw, cin)              # [cout][trow][kx][ci]
    full = torch.zeros(M, kt * kh, kw, cpad, dtype=torch.float32, device=w.device)
    ridx = torch.tensor(rows, device=w.device)
    src = dst = 0
    for c, p in zip(seg_channels, seg_padded):
        full[ridx, :, :, dst:dst + c] = wk[:, :, :, src:src + c]
        src += c
        dst += p
    nk = kt * kh * nchunk * kw
    # [blk][mb][r][trow][kx][chunk][h][j] -> [trow][chunk][kx][blk][mb][h][r][j]
    t = full.reshape(M // 64, 2, 32, kt * kh, kw, nchunk, 2, 8).permute(3, 5, 4, 0, 1, 6, 2, 7).contiguous()
    t = t.reshape(nk, M // 64, 2, 64, 8)
    hi, lo = split_bf16(t)
    packed = torch.stack([hi, lo], dim=3).contiguous()                    # [ks][blk][mb][plane][lane][8]
    b = torch.zeros(M, dtype=torch.float32, device=w.device)
    if bias is not None:
        b[ridx] = bias.detach().float()
    meta = dict(M=M, nk=nk, taps=(kt, kh, kw), cpad=cpad, seg_padded=seg_padded, version=4)
    return packed.reshape(-1), b, meta


def pack_conv6(weight: torch.Tensor, bias: Optional[torch.Tensor], seg_channels: Sequence[int],
               seg_padded: Optional[Sequence[int]] = None, cout_map: Optional[Sequence[int]] = None,
               m_pad: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, dict]:
    """Layout of the one-wave-per-SIMD large-map kernel (ppmstereo_amd/csrc/conv_gemm6.hip, v_mfma_f32_16x16x32_bf16): the A-operand images
    of 16 couts x 32 input channels, so that a wave loads its weight fragments straight from L2 into registers with 1 KiB loads:
        bf16 [k32-step][M/16][plane (hi, lo)][lane = 16*kg + r][8] = W[cout = 16*blk + r][cin = 32*chunk + 8*kg + j] of tap (row-step, s),
        k32-step = ((kz*kh + ky) * nchunk32 + chunk) * kw + kx.
    As for pack_conv4 the LAST kernel axis is the swept one: y-swept convs are passed with kh / kw swapped, 2-D swept ones with (ky, kx)
    flattened into x; the input segments are concatenated along K, each zero-padded to a multiple of 32 channels."""
    w = weight.detach().float()
    if w.dim() == 4:
        w = w[:, :, None]
    cout, cin, kt, kh, kw = w.shape
    assert sum(seg_channels) == cin, (seg_channels, cin)
    seg_padded = [_pad_to(c, 32) for c in seg_channels] if seg_padded is None else list(seg_padded)
    assert all(p % 32 == 0 and p >= c for p, c in zip(seg_padded, seg_channels))
    cpad = sum(seg_padded)
    nchunk = cpad // 32
    rows = list(range(cout)) if cout_map is None else list(cout_map)
    M = _pad_to(max(rows) + 1, 64) if m_pad is None else m_pad
    assert M % 16 == 0 and max(rows) < M
    wk = w.permute(0, 2, 3, 4, 1).reshape(cout, kt * kh, kw, cin)              # [cout][trow][kx][ci]
    full = torch.zeros(M, kt * kh, kw, cpad, dtype=torch.float32, device=w.device)
    ridx = torch.tensor(rows, device=w.device)
    src = dst = 0
    for c, p in zip(seg_channels, seg_padded):
        full[ridx, :, :, dst:dst + c] = wk[:, :, :, src:src + c]
        src += c
        dst += p
    nk = kt * kh * nchunk * kw
    # [blk][r][trow][kx][chunk][kg][j] -> [trow][chunk][kx][blk][kg][r][j]
    t = full.reshape(M // 16, 16, kt * kh, kw, nchunk, 4, 8).permute(2, 4, 3, 0, 5, 1, 6).contiguous()
    t = t.reshape(nk, M // 16, 64, 8)
    hi, lo = split_bf16(t)
    packed = torch.stack([hi, lo], dim=2).contiguous()                    # [ks][blk][plane][lane][8]
    b = torch.zeros(M, dtype=torch.float32, device=w.device)
    if bias is not None:
        b[ridx] = bias.detach().float()
    meta = dict(M=M, nk=nk, taps=(kt, kh, kw), cpad=cpad, seg_padded=seg_padded, version=8)
    return packed.reshape(-1), b, meta


def pack_conv6_grouped(weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], group_channels: int) -> Tuple[torch.Tensor, torch.Tensor, dict]:
    """Two convolutions of the same shape (cout = 128, cin = group_channels, the same taps) as ONE grouped launch of conv_gemm6 (ppms_conv.groups == 2:
    input segment g feeds the couts of epilogue half g): each is packed as pack_conv6 packs an M = 128 convolution, and the two images are interleaved per
    k32-step -- [k32-step][16 cout blocks: 8 of group 0, 8 of group 1][plane][lane][8] -- which is the M = 256 layout the kernel's four waves read
    (wave w: blocks 4 w .. 4 w + 3, i.e. waves 0-1 group 0, waves 2-3 group 1)."""
    assert len(weights) == 2 and len(biases) == 2
    parts, bs, meta0 = [], [], None
    for w, b in zip(weights, biases):
        assert w.shape[0] <= 128 and w.shape[1] == group_channels, (tuple(w.shape), group_channels)
        packed, bias, meta = pack_conv6(w, b, [group_channels], None, None, 128)
        parts.append(packed.reshape(meta["nk"], 8, 2, 64, 8))
        bs.append(bias)
        assert meta0 is None or (meta0["nk"], meta0["taps"], meta0["cpad"]) == (meta["nk"], meta["taps"], meta["cpad"])
        meta0 = meta
    packed = torch.cat(parts, dim=1).contiguous()                          # [ks][16 blocks][plane][lane][8]
    meta = dict(M=256, nk=meta0["nk"], taps=meta0["taps"], cpad=meta0["cpad"], seg_padded=[meta0["cpad"], meta0["cpad"]], version=8, groups=2)
    return packed.reshape(-1), torch.cat(bs), meta


def unpack_conv6_reference(packed: torch.Tensor, M: int, nk: int, taps, nchunk: int) -> torch.Tensor:
    """Inverse of pack_conv6 -> fp32 [M][K] in the plain K order (k = tap*Cpad + ci), for the host-logic tests."""
    kt, kh, kw = taps
    t = packed.reshape(nk, M // 16, 2, 64, 8).float()                      # [ks][blk][plane][lane][8]
    t = t[:, :, 0] + t[:, :, 1]                                            # [ks][blk][lane][8]
    t = t.reshape(kt * kh, nchunk, kw, M // 16, 4, 16, 8)                  # [trow][chunk][kx][blk][kg][r][j]
    t = t.permute(3, 5, 0, 2, 1, 4, 6)                                     # [blk][r][trow][kx][chunk][kg][j]
    return t.reshape(M, kt * kh * kw * nchunk * 32)


def pack_gemm1(weight: torch.Tensor, bias: Optional[torch.Tensor], seg_channels: Sequence[int],
               seg_padded: Optional[Sequence[int]] = None, cout_map: Optional[Sequence[int]] = None,
               m_pad: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, dict]:
    """Layout of the thin-GEMM kernel for 1x1 convolutions (ppmstereo_amd/csrc/gemm1.hip): per (32-cout block, k16-step, plane) the
    1 KiB MFMA A-operand image, so that a wave loads every fragment of its K slice with one 16-byte request per lane:
        bf16 [M/32][K/16][plane (hi, lo)][lane = 32*h + r][8] = W[cout = 32*blk + r][k = 16*step + 8*h + j].
    The input segments are concatenated along K, each zero-padded to a multiple of 16 channels."""
    w = weight.detach().float()
    w = w.reshape(w.shape[0], w.shape[1], -1)
    assert w.shape[2] == 1, "pack_gemm1: 1x1 kernels only"
    w = w[:, :, 0]
    cout, cin = w.shape
    assert sum(seg_channels) == cin, (seg_channels, cin)
    seg_padded = [_pad_to(c, 16) for c in seg_channels] if seg_padded is None else list(seg_padded)
    assert all(p % 16 == 0 and p >= c for p, c in zip(seg_padded, seg_channels))
    cpad = sum(seg_padded)
    rows = list(range(cout)) if cout_map is None else list(cout_map)
    M = _pad_to(max(rows) + 1, 32) if m_pad is None else m_pad
    assert M % 32 == 0 and max(rows) < M
    full = torch.zeros(M, cpad, dtype=torch.float32, device=w.device)
    ridx = torch.tensor(rows, device=w.device)
    src = dst = 0
    for c, p in zip(seg_channels, seg_padded):
        full[ridx, dst:dst + c] = w[:, src:src + c]
        src += c
        dst += p
    nk = cpad // 16
    t = full.reshape(M // 32, 32, nk, 2, 8).permute(0, 2, 3, 1, 4).contiguous()          # [blk][step][h][r][j]
    t = t.reshape(M // 32, nk, 64, 8)
    hi, lo = split_bf16(t)
    packed = torch.stack([hi, lo], dim=2).contiguous()                                   # [blk][step][plane][lane][8]
    b = torch.zeros(M, dtype=torch.float32, device=w.device)
    if bias is not None:
        b[ridx] = bias.detach().float()
    meta = dict(M=M, nk=nk, taps=(1, 1, 1), cpad=cpad, seg_padded=seg_padded, version=6)
    return packed.reshape(-1), b, meta


def pack_stream(weight: torch.Tensor, bias: Optional[torch.Tensor], seg_channels: Sequence[int],
                seg_padded: Optional[Sequence[int]] = None, cout_map: Optional[Sequence[int]] = None,
                m_pad: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, dict]:
    """Layout of the register-streamed small-map kernel (ppmstereo_amd/csrc/conv_stream.hip): pack_gemm1's A-operand images with taps,
        bf16 [M/32][tap][K/16][plane (hi, lo)][lane = 32*h + r][8] = W[cout = 32*blk + r][tap][k = 16*chunk + 8*h + j],
    tap = (kz*kh + ky)*kw + kx in the natural order (no sweep), the input segments concatenated along K, each zero-padded to a multiple of
    16 channels; the padded K must be a multiple of 64 (every tap's chunks are dealt to the four waves of a workgroup in equal shares)."""
    w = weight.detach().float()
    if w.dim() == 4:
        w = w[:, :, None]
    cout, cin, kt, kh, kw = w.shape
    assert sum(seg_channels) == cin, (seg_channels, cin)
    seg_padded = [_pad_to(c, 16) for c in seg_channels] if seg_padded is None else list(seg_padded)
    assert all(p % 16 == 0 and p >= c for p, c in zip(seg_padded, seg_channels))
    cpad = sum(seg_padded)
    assert cpad % 64 == 0, f"pack_stream: padded input channels {cpad} must be a multiple of 64"
    rows = list(range(cout)) if cout_map is None else list(cout_map)
    M = _pad_to(max(rows) + 1, 64) if m_pad is None else m_pad
    assert M % 64 == 0 and max(rows) < M
    taps = kt * kh * kw
    wk = w.permute(0, 2, 3, 4, 1).reshape(cout, taps, cin)                     # [cout][tap][ci]
    full = torch.zeros(M, taps, cpad, dtype=torch.float32, device=w.device)
    ridx = torch.tensor(rows, device=w.device)
    src = dst = 0
    for c, p in zip(seg_channels, seg_padded):
        full[ridx, :, dst:dst + c] = wk[:, :, src:src + c]
        src += c
        dst += p
    nk16 = cpad // 16
    t = full.reshape(M // 32, 32, taps, nk16, 2, 8).permute(0, 2, 3, 4, 1, 5).contiguous()          # [blk][tap][chunk][h][r][j]
    t = t.reshape(M // 32, taps * nk16, 64, 8)
    hi, lo = split_bf16(t)
    packed = torch.stack([hi, lo], dim=2).contiguous()                                               # [blk][step][plane][lane][8]
    b = torch.zeros(M, dtype=torch.float32, device=w.device)
    if bias is not None:
        b[ridx] = bias.detach().float()
    meta = dict(M=M, nk=taps * nk16, taps=(kt, kh, kw), cpad=cpad, seg_padded=seg_padded, version=7)
    return packed.reshape(-1), b, meta


def unpack_stream_reference(packed: torch.Tensor, M: int, taps: int, nk16: int) -> torch.Tensor:
    """Inverse of pack_stream -> fp32 [M][K] in the plain K order (k = tap*Cpad + ci), for the host-logic tests."""
    t = packed.reshape(M // 32, taps, nk16, 2, 2, 32, 8).float()                                     # [blk][tap][chunk][plane][h][r][j]
    t = t[:, :, :, 0] + t[:, :, :, 1]                                                                 # [blk][tap][chunk][h][r][j]
    return t.permute(0, 4, 1, 2, 3, 5).reshape(M, taps * nk16 * 16)


def unpack_gemm1_reference(packed: torch.Tensor, M: int, nk: int) -> torch.Tensor:
    """Inverse of pack_gemm1 -> fp32 [M][K], for the host-logic tests."""
    t = packed.reshape(M // 32, nk, 2, 2, 32, 8).float()                                 # [blk][step][plane][h][r][j]
    t = t[:, :, 0] + t[:, :, 1]                                                           # [blk][step][h][r][j]
    return t.permute(0, 3, 1, 2, 4).reshape(M, nk * 16)


def unpack_conv4_reference(packed: torch.Tensor, M: int, nk: int, taps, nchunk: int) -> torch.Tensor:
    """Inverse of pack_conv4 -> fp32 [M][K] in the plain K order (k = tap*Cpad + ci), for the host-logic tests."""
    kt, kh, kw = taps
    t = packed.reshape(nk, M // 64, 2, 2, 64, 8).float()                  # [ks][blk][mb][plane][lane][8]
    t = t[:, :, :, 0] + t[:, :, :, 1]                                      # [ks][blk][mb][lane][8]
    t = t.reshape(kt * kh, nchunk, kw, M // 64, 2, 2, 32, 8)               # [trow][chunk][kx][blk][mb][h][r][j]
    t = t.permute(3, 4, 6, 0, 2, 1, 5, 7)                                  # [blk][mb][r][trow][kx][chunk][h][j]
    return t.reshape(M, kt * kh * kw * nchunk * 16)


def unpack_conv2_reference(packed: torch.Tensor, M: int, nk: int, taps, nchunk: int) -> torch.Tensor:
    """Inverse of pack_conv2 -> fp32 [M][K] in the plain K order (k = tap*Cpad + ci, tap = (kz*kh + ky)*kw + kx), for the host-logic tests."""
    kt, kh, kw = taps
    t = packed.reshape(nk, M // BM, 2, BM, 4, 8).float()
    t = t[:, :, 0] + t[:, :, 1]
    m = torch.arange(BM)
    chunk = torch.arange(4)
    pos = chunk[None, :] ^ ((m[:, None] >> 2) & 3)
    un = t[:, :, m[:, None].expand(BM, 4), pos]                            # [ks][mblk][m][c][8]
    un = un.reshape(kt * kh, nchunk, kw, M // BM, BM, 32).permute(3, 4, 0, 2, 1, 5)      # [mblk][m][trow][kx][chunk][32]
    return un.reshape(M, kt * kh * kw * nchunk * 32)
