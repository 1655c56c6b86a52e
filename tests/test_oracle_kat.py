"""Known-answer tests for the restated third-party blocks (croco / must3r; parity unpinned -- SURVEY 8(c)).

No reference vectors exist for these, so they are checked against mathematical identities and against independent
formulations (torch SDPA, naive complex rotation)."""
import math
import torch
import torch.nn.functional as F

from oracle.blocks import RoPE2D, Attention, Block, CrossAttention, _sdpa
from oracle import must3r as OM
from panst3r_amd.synthetic import fill_module_


def test_rope_identities():
    rope = RoPE2D(100.0)
    B, H, N, hd = 2, 3, 12, 64
    t = torch.randn(B, H, N, hd)
    pos = torch.stack([torch.randint(0, 24, (B, N)), torch.randint(0, 32, (B, N))], -1)
    out = rope(t, pos)
    assert torch.allclose(out.norm(dim=-1), t.norm(dim=-1), atol=1e-4)                 # rotations preserve the norm
    assert torch.allclose(rope(t, torch.zeros_like(pos)), t, atol=1e-6)                # position 0 is the identity
    # relative property: <R(p)q, R(p')k> depends on p - p' only
    q, k = torch.randn(1, 1, 1, hd), torch.randn(1, 1, 1, hd)
    def dot(pq, pk):
        return (rope(q, torch.tensor([[pq]])) * rope(k, torch.tensor([[pk]]))).sum()
    assert abs(dot([3, 5], [1, 2]) - dot([13, 25], [11, 22])) < 1e-4
    # against a naive complex rotation of channel pairs (i, i + D/2) with freq 100^(-2i/D), D = hd/2
    D = hd // 2
    inv = 100.0 ** (-torch.arange(0, D, 2).float() / D)
    ref = t.clone()
    for half, col in ((0, 0), (1, 1)):
        x = t[..., half * D:(half + 1) * D]
        ang = pos[..., col][:, None, :, None].float() * inv
        a, b = x[..., :D // 2], x[..., D // 2:]
        ref[..., half * D:(half + 1) * D] = torch.cat([a * ang.cos() - b * ang.sin(), b * ang.cos() + a * ang.sin()], -1)
    assert torch.allclose(out, ref, atol=1e-5)


def test_attention_matches_sdpa():
    q, k, v = torch.randn(2, 4, 10, 16), torch.randn(2, 4, 13, 16), torch.randn(2, 4, 13, 16)
    assert torch.allclose(_sdpa(q, k, v), F.scaled_dot_product_attention(q, k, v), atol=1e-5)
    blk = Block(32, 4, qkv_bias=True).eval()
    x = torch.randn(2, 6, 32)
    with torch.no_grad():
        y = blk(x, None)
        h = blk.norm1(x)
        qkv = blk.attn.qkv(h).reshape(2, 6, 3, 4, 8).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(2, 6, 32)
        x1 = x + blk.attn.proj(a)
        ref = x1 + blk.mlp(blk.norm2(x1))
    assert torch.allclose(y, ref, atol=1e-5)


def _tiny_decoder():
    return fill_module_(OM.MUSt3R(img_size=[64, 64], enc_embed_dim=32, embed_dim=32, depth=2, num_heads=2).eval(), seed=3)


@torch.no_grad()
def test_must3r_memory_invariants():
    dec = _tiny_decoder()
    T, H, W = 6, 32, 48
    ys, xs = torch.meshgrid(torch.arange(2), torch.arange(3), indexing='ij')
    pos = torch.stack([ys, xs], -1).reshape(1, 1, T, 2)
    x = torch.randn(1, 3, T, 32)
    ts = torch.tensor([[[H, W]] * 3])
    mem, pm, feats = dec(x[:, :2], pos.expand(1, 2, -1, -1), ts[:, :2], None, render=False, return_feats=True)
    assert [m.shape for m in mem[0]] == [(1, 2 * T, 32)] * 2 and mem[2] == 2
    assert mem[1].tolist() == [[0] * T + [1] * T]
    assert pm.shape == (1, 2, H, W, 7) and feats[-1].shape == (1, 2, T, 32)
    mem2, _, _ = dec(x[:, 2:], pos, ts[:, 2:], mem, render=False, return_feats=True)
    assert mem2[0][0].shape == (1, 3 * T, 32) and torch.equal(mem2[0][0][:, :2 * T], mem[0][0])     # append-only
    # rendering does not touch the memory and is independent per view (batched == one by one)
    mem3, pm_all, f_all = dec(x, pos.expand(1, 3, -1, -1), ts, mem2, render=True, return_feats=True)
    assert all(torch.equal(a, b) for a, b in zip(mem3[0], mem2[0]))
    for i in range(3):
        _, pm_i, f_i = dec(x[:, i:i + 1], pos, ts[:, i:i + 1], mem2, render=True, return_feats=True)
        assert torch.allclose(pm_i[:, 0], pm_all[:, i], atol=1e-5) and torch.allclose(f_i[-1][:, 0], f_all[-1][:, i], atol=1e-5)
    # the reference image carries no image2_embed in update mode
    dec.image2_embed.data.fill_(5.0)
    _, pm_b, _ = dec(x[:, :2], pos.expand(1, 2, -1, -1), ts[:, :2], None, render=False, return_feats=True)
    assert not torch.allclose(pm_b[:, 1], pm[:, 1], atol=1e-3)


@torch.no_grad()
def test_pointmap_head_is_pixel_shuffle():
    head = OM.LinearHead(8, 4, 7)
    tok = torch.randn(1, 6, 8)
    out = head(tok, 8, 12)                               # 2x3 tokens of 4x4 pixels
    full = head.proj(tok)[0].reshape(2, 3, 7, 4, 4)      # [ty, tx, c, dy, dx]
    assert torch.allclose(out[0, 5, 9], full[1, 2, :, 1, 1], atol=1e-6)


def test_mem_batches_and_encoder_positions():
    assert OM.mem_batches_for(2) == [2] and OM.mem_batches_for(5) == [2, 1, 1, 1]
    enc = OM.Dust3rEncoder(img_size=[64, 64], embed_dim=32, depth=1, num_heads=2).eval()
    with torch.no_grad():
        x, pos = enc(torch.randn(2, 3, 32, 48), None)
    assert x.shape == (2, 6, 32) and pos[0].tolist() == [[0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [1, 2]]


def test_block_equals_an_independent_vit_layer():
    """The restated croco `Block` (pre-LN: x += attn(norm1 x); x += mlp(norm2 x); fused qkv in (q, k, v) order, head-major channel split, exact-erf GELU) has
    no upstream source here to be pinned against - but without RoPE it IS the standard ViT encoder layer, and the installed `transformers` carries an
    independent implementation of that (ViTLayer: separate q / k / v projections, its own attention and MLP code).  Same weights mapped across -> same outputs.
    This pins the block's structure (norm placement, residual order, qkv row order, head split, softmax scale, activation) to a second implementation; the
    RoPE-2D rotation itself stays on test_rope_identities' naive complex form."""
    from transformers import ViTConfig
    from transformers.models.vit.modeling_vit import ViTLayer
    torch.manual_seed(0)
    D, H = 64, 4
    blk = Block(D, H, mlp_ratio=4.0, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6)).eval()
    fill_module_(blk, seed=5)
    cfg = ViTConfig(hidden_size=D, num_attention_heads=H, intermediate_size=4 * D, hidden_act='gelu', layer_norm_eps=1e-6, attention_probs_dropout_prob=0.0,
                    hidden_dropout_prob=0.0, qkv_bias=True)
    cfg._attn_implementation = 'eager'
    ref = ViTLayer(cfg).eval()
    sd = blk.state_dict()
    qw, kw, vw = sd['attn.qkv.weight'].chunk(3, 0)
    qb, kb, vb = sd['attn.qkv.bias'].chunk(3, 0)
    ref.load_state_dict({'attention.q_proj.weight': qw, 'attention.q_proj.bias': qb, 'attention.k_proj.weight': kw, 'attention.k_proj.bias': kb,
                         'attention.v_proj.weight': vw, 'attention.v_proj.bias': vb, 'attention.o_proj.weight': sd['attn.proj.weight'],
                         'attention.o_proj.bias': sd['attn.proj.bias'], 'layernorm_before.weight': sd['norm1.weight'], 'layernorm_before.bias': sd['norm1.bias'],
                         'layernorm_after.weight': sd['norm2.weight'], 'layernorm_after.bias': sd['norm2.bias'], 'mlp.fc1.weight': sd['mlp.fc1.weight'],
                         'mlp.fc1.bias': sd['mlp.fc1.bias'], 'mlp.fc2.weight': sd['mlp.fc2.weight'], 'mlp.fc2.bias': sd['mlp.fc2.bias']}, strict=True)
    x = torch.randn(2, 11, D) * 1.5
    with torch.no_grad():
        got = blk(x, None)
        want = ref(x)
        want = want[0] if isinstance(want, (tuple, list)) else want
    assert torch.allclose(got, want, atol=2e-6, rtol=1e-5), float((got - want).abs().max())


def test_cross_attention_equals_torch_multi_head_attention():
    """The restated croco `CrossAttention` (separate projq / projk / projv, head-major channel split, output projection) against torch's own multi-head attention
    with separate projection weights (F.multi_head_attention_forward, use_separate_proj_weight=True): an independent implementation of the same operator."""
    torch.manual_seed(1)
    D, H, B, Nq, Nk = 48, 4, 2, 7, 11
    ca = CrossAttention(D, rope=None, num_heads=H, qkv_bias=True).eval()
    fill_module_(ca, seed=9)
    x, y = torch.randn(B, Nq, D), torch.randn(B, Nk, D)
    with torch.no_grad():
        got = ca(x, y, y, None, None)
        want, _ = F.multi_head_attention_forward(
            x.transpose(0, 1), y.transpose(0, 1), y.transpose(0, 1), D, H, None, torch.cat([ca.projq.bias, ca.projk.bias, ca.projv.bias]), None, None, False, 0.0,
            ca.proj.weight, ca.proj.bias, training=False, need_weights=False, use_separate_proj_weight=True, q_proj_weight=ca.projq.weight,
            k_proj_weight=ca.projk.weight, v_proj_weight=ca.projv.weight)
    assert torch.allclose(got, want.transpose(0, 1), atol=2e-6, rtol=1e-5), float((got - want.transpose(0, 1)).abs().max())
