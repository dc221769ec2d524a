"""Independent cross-check of the restated DINOv2 (oracle/dinov2_ref.py) against the
``transformers`` implementation, through the facebookresearch -> HF weight-key remap of
SURVEY.md appendix C.  Compared at 518x518 (the native grid: neither side interpolates the
positional table, the one place where HF deliberately differs from the hub code)."""
import pytest
import torch

from anyloc_amd import synth
from oracle import dinov2_ref

transformers = pytest.importorskip("transformers")


def to_hf(sd, depth, swiglu):
    out = {"embeddings.cls_token": sd["cls_token"], "embeddings.mask_token": sd["mask_token"],
           "embeddings.position_embeddings": sd["pos_embed"],
           "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
           "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
           "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    for i in range(depth):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        D = sd[p + "attn.proj.weight"].shape[0]
        for j, name in enumerate(("query", "key", "value")):
            out[q + f"attention.attention.{name}.weight"] = sd[p + "attn.qkv.weight"][j * D:(j + 1) * D]
            out[q + f"attention.attention.{name}.bias"] = sd[p + "attn.qkv.bias"][j * D:(j + 1) * D]
        out[q + "attention.output.dense.weight"] = sd[p + "attn.proj.weight"]
        out[q + "attention.output.dense.bias"] = sd[p + "attn.proj.bias"]
        for a, b in (("norm1", "norm1"), ("norm2", "norm2")):
            out[q + b + ".weight"], out[q + b + ".bias"] = sd[p + a + ".weight"], sd[p + a + ".bias"]
        out[q + "layer_scale1.lambda1"], out[q + "layer_scale2.lambda1"] = sd[p + "ls1.gamma"], sd[p + "ls2.gamma"]
        if swiglu:
            out[q + "mlp.weights_in.weight"], out[q + "mlp.weights_in.bias"] = sd[p + "mlp.w12.weight"], sd[p + "mlp.w12.bias"]
            out[q + "mlp.weights_out.weight"], out[q + "mlp.weights_out.bias"] = sd[p + "mlp.w3.weight"], sd[p + "mlp.w3.bias"]
        else:
            for f in ("fc1", "fc2"):
                out[q + f"mlp.{f}.weight"], out[q + f"mlp.{f}.bias"] = sd[p + f"mlp.{f}.weight"], sd[p + f"mlp.{f}.bias"]
    return out


@pytest.mark.parametrize("name,depth", [("dinov2_vits14", 4), ("dinov2_vitg14", 1)])
def test_restatement_matches_hf(name, depth):
    dim, _, heads, ffn, hidden = dinov2_ref.ARCH[name]
    sd = synth.synthetic_state_dict(name, 3, depth=depth)
    ours = dinov2_ref.DinoVisionTransformer(name)
    ours.blocks = ours.blocks[:depth]
    ours.load_state_dict(sd, strict=True)
    ours.eval()
    cfg = transformers.Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                                    mlp_ratio=4, image_size=518, patch_size=14, layerscale_value=1.0,
                                    use_swiglu_ffn=(ffn == "swiglu"), layer_norm_eps=1e-6, qkv_bias=True,
                                    hidden_act="gelu", attn_implementation="eager")
    hf = transformers.Dinov2Model(cfg).eval()
    missing, unexpected = hf.load_state_dict(to_hf(sd, depth, ffn == "swiglu"), strict=False)
    assert not unexpected and all("mask" in m or "pooler" in m for m in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, 3, 518, 518, generator=g)
    with torch.no_grad():
        x = ours.prepare_tokens(img)
        hid = [x]
        for blk in ours.blocks:
            x = blk(x)
            hid.append(x)
        y = ours.norm(x)
        ref = hf(pixel_values=img, output_hidden_states=True)
    for a, b in zip(hid, ref.hidden_states):
        assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(b.abs().max()))
    assert float((y - ref.last_hidden_state).abs().max()) < 2e-4
    # the 'value' facet the reference hooks == HF's separate value projection of the normed input
    with torch.no_grad():
        v_ours = ours.blocks[0].attn.qkv(ours.blocks[0].norm1(hid[0]))[..., 2 * dim:]
        lay = hf.encoder.layer[0]
        v_hf = lay.attention.attention.value(lay.norm1(ref.hidden_states[0]))
    assert float((v_ours - v_hf).abs().max()) < 1e-4
