"""Pins the gfx950 hardware layouts the kernels rely on (MFMA operand/result maps and the
ds_read_b64_tr_b16 transpose read).  Raw tables are dumped to gpurun_out/probe_layouts.json."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf16_bits(x):
    return x.to(torch.bfloat16)


def test_mfma_16x16x32_layout(out_dir):
    from mmf_amd import _native as nat
    dev = "cuda"
    lane = torch.arange(64)
    # A: lane l supplies tile row (l & 15); B: lane l supplies tile column (l & 15).
    a = torch.zeros(64, 8); b = torch.zeros(64, 8)
    a[:, :] = ((lane & 15) + 1).float()[:, None]
    b[:, :] = (2.0 ** (lane & 15).float())[:, None]
    d = torch.zeros(64, 4, device=dev)
    nat.probe_mfma16(_bf16_bits(a).to(dev), _bf16_bits(b).to(dev), d)
    torch.cuda.synchronize()
    d = d.cpu()
    exp = torch.zeros(64, 4)
    for l in range(64):
        for r in range(4):
            row, col = (l >> 4) * 4 + r, l & 15
            exp[l, r] = 32.0 * (row + 1) * 2.0 ** col
    json.dump({"d": d.tolist()}, open(os.path.join(out_dir, "probe_mfma16.json"), "w"))
    assert torch.equal(d, exp), "C/D map of v_mfma_f32_16x16x32_bf16 differs from (row=(l>>4)*4+r, col=l&15)"


def test_mfma_16x16x32_slot_pairing():
    """slot (g, e) of the A operand multiplies slot (g, e) of the B operand and nothing else."""
    from mmf_amd import _native as nat
    dev = "cuda"
    got = torch.zeros(32, 32)
    a = torch.zeros(64, 8); b = torch.zeros(64, 8)
    for ga in range(4):
        for ea in range(8):
            a.zero_()
            a[ga * 16:(ga + 1) * 16, ea] = 1.0  # every row, one slot
            b.zero_()
            for gb in range(4):
                for eb in range(8):
                    b[gb * 16:(gb + 1) * 16, eb] = float(gb * 8 + eb + 1)
            d = torch.zeros(64, 4, device=dev)
            nat.probe_mfma16(_bf16_bits(a).to(dev), _bf16_bits(b).to(dev), d)
            v = d.cpu()[0, 0].item()
            assert v == float(ga * 8 + ea + 1), (ga, ea, v)


def test_mfma_32x32x16_layout(out_dir):
    from mmf_amd import _native as nat
    dev = "cuda"
    lane = torch.arange(64)
    a = torch.zeros(64, 8); b = torch.zeros(64, 8)
    a[:, :] = ((lane & 31) + 1).float()[:, None]
    b[:, :] = (2.0 ** (lane & 31).float())[:, None]
    d = torch.zeros(64, 16, device=dev)
    nat.probe_mfma32(_bf16_bits(a).to(dev), _bf16_bits(b).to(dev), d)
    torch.cuda.synchronize()
    d = d.cpu()
    exp = torch.zeros(64, 16)
    for l in range(64):
        for r in range(16):
            row, col = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31
            exp[l, r] = 16.0 * (row + 1) * 2.0 ** col
    json.dump({"d": d.tolist()}, open(os.path.join(out_dir, "probe_mfma32.json"), "w"))
    assert torch.equal(d, exp), "C/D map of v_mfma_f32_32x32x16_bf16 differs from the documented one"


def test_mfma_32x32x16_slot_pairing():
    from mmf_amd import _native as nat
    dev = "cuda"
    a = torch.zeros(64, 8); b = torch.zeros(64, 8)
    for ha in range(2):
        for ea in range(8):
            a.zero_(); a[ha * 32:(ha + 1) * 32, ea] = 1.0
            b.zero_()
            for hb in range(2):
                for eb in range(8):
                    b[hb * 32:(hb + 1) * 32, eb] = float(hb * 8 + eb + 1)
            d = torch.zeros(64, 16, device=dev)
            nat.probe_mfma32(_bf16_bits(a).to(dev), _bf16_bits(b).to(dev), d)
            v = d.cpu()[0, 0].item()
            assert v == float(ha * 8 + ea + 1), (ha, ea, v)


def test_ds_read_tr16_b64_semantics(out_dir):
    """result[lane i of a 16-lane group][j] = element (i % 4) of what lane (4 j + i / 4) of the same
    group addressed: a 16-lane group reads a 4 x 16 block (row = lane / 4, 8 bytes per lane) and
    lane i receives column i."""
    from mmf_amd import _native as nat
    dev = "cuda"
    img = torch.arange(2048, dtype=torch.int16)
    tables = {}
    for name, addr in (
        ("contiguous", torch.arange(64, dtype=torch.int32) * 8),
        # rows of 256 bytes: lane p of group g reads row (8 g + p / 4), bytes (p % 4) * 8 (GEMM k-major image, rot = 0)
        ("rows256", torch.tensor([((8 * (l >> 4) + ((l & 15) >> 2)) * 256 + (l & 3) * 8) % 4096 for l in range(64)], dtype=torch.int32)),
    ):
        out = torch.zeros(64, 4, dtype=torch.int16, device=dev)
        nat.probe_tr16(img.to(dev), addr.to(dev), out)
        torch.cuda.synchronize()
        out = out.cpu()
        tables[name] = out.tolist()
        exp = torch.zeros(64, 4, dtype=torch.int16)
        for l in range(64):
            g, i = l >> 4, l & 15
            for j in range(4):
                src_lane = g * 16 + 4 * j + (i >> 2)
                exp[l, j] = addr[src_lane].item() // 2 + (i & 3)
        tables[name + "_expected"] = exp.tolist()
        json.dump(tables, open(os.path.join(out_dir, "probe_tr16.json"), "w"))
        assert torch.equal(out, exp), "ds_read_b64_tr_b16 semantics differ (%s); see gpurun_out/probe_tr16.json" % name
