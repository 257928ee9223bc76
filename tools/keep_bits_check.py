"""Development aid: the attention forward's dropout keep-bit table (mmf_attn_desc.keep_bits) against a host restatement of the counter hash."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmf_amd import _native as nat

M32 = 0xFFFFFFFF


def mix24(x):
    x = x & M32
    x = x ^ (x >> 16)
    x = (((x & 0xFFFFFF) * 0xB5297B) + (((x << 9) | (x >> 23)) & M32)) & M32
    x = x ^ (x >> 13)
    x = (((x & 0xFFFFFF) * 0x68E31F) + (((x << 11) | (x >> 21)) & M32)) & M32
    x = x ^ (x >> 15)
    return x


def expected_keep(B, heads, S, key, thr16, dev):
    skp = (S + 31) // 32 * 32
    bh = torch.arange(B * heads, device=dev, dtype=torch.int64)[:, None, None]
    q = torch.arange(S, device=dev, dtype=torch.int64)[None, :, None]
    k = torch.arange(skp, device=dev, dtype=torch.int64)[None, None, :]
    idx = ((bh * S + q) * skp + k) & M32
    h = mix24((idx >> 1) + key)
    half = torch.where((k & 1) == 1, h >> 16, h & 0xFFFF)
    return half >= thr16          # [BH, S, skp]


if __name__ == "__main__":
    B, heads, S, d = 2, 3, 228, 64
    H = heads * d
    dev = "cuda"
    qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).bfloat16()
    drop = nat.drop_cfg(0.1, 424242)
    words = nat.attention_keep_bits_words(B, heads, S, S, d)
    kb = torch.zeros(words, dtype=torch.int32, device=dev)
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, heads, S, device=dev)
    nat.attention_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, None, ctx, H, lse, B, heads, S, S, 0.125, drop, keep_bits=kb)
    torch.cuda.synchronize()
    nt = (S + 31) // 32
    tab = kb.view(B * heads, nt, nt, 32).long() & M32          # [bh, qt, kt, key j] bit x = query
    bits = torch.stack([(tab >> x) & 1 for x in range(32)], dim=-1)      # [bh, qt, kt, j, x]
    got = bits.permute(0, 1, 4, 2, 3).reshape(B * heads, nt * 32, nt * 32).bool()      # [bh, q, key]
    exp = expected_keep(B, heads, S, drop[0], drop[1], dev)
    g, e = got[:, :S, :S], exp[:, :S, :S]
    bad = g != e
    print("keep rate got %.4f expected %.4f, mismatches %d of %d" % (g.float().mean(), e.float().mean(), int(bad.sum()), bad.numel()))
    if bad.any():
        ix = torch.nonzero(bad)
        print("first mismatches (bh, q, key):", ix[:12].tolist())
        qq, kk = ix[:, 1], ix[:, 2]
        print("by q%32:", torch.bincount(qq % 32, minlength=32).tolist())
        print("by key%32:", torch.bincount(kk % 32, minlength=32).tolist())
        print("by q tile:", torch.bincount(qq // 32, minlength=nt).tolist(), " by key tile:", torch.bincount(kk // 32, minlength=nt).tolist())
