"""Test helper: HuggingFace ViTModel (an independent pre-LN ViT implementation) loaded with M3AE-encoder
weights, to pin oracle/m3ae_np.py.  Not part of the product."""
import numpy as np
import torch


def build_hf_vit(P, cfg, pos):
    from transformers import ViTConfig, ViTModel
    D, Pp = cfg.width, cfg.patch
    c = ViTConfig(hidden_size=D, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, intermediate_size=cfg.mlp_ratio * D,
                  hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, image_size=cfg.img_res, patch_size=Pp, num_channels=3,
                  qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager")
    m = ViTModel(c, add_pooling_layer=False).double().eval()
    t = lambda a: torch.from_numpy(np.asarray(a, np.float64).copy())
    sd = {"embeddings.cls_token": t(P["cls_token"])}
    pe = np.zeros((1, cfg.tokens, D))
    pe[0, 1:] = pos + np.asarray(P["encoder_image_type_embedding"], np.float64)[0]
    sd["embeddings.position_embeddings"] = t(pe)
    k = np.asarray(P["image_embedding/kernel"], np.float64).reshape(Pp, Pp, 3, D)  # (p1, p2, c, d)
    sd["embeddings.patch_embeddings.projection.weight"] = t(k.transpose(3, 2, 0, 1))
    sd["embeddings.patch_embeddings.projection.bias"] = t(P["image_embedding/bias"])
    for i in range(cfg.layers):
        s, o = f"encoder/Block_{i}/", f"layers.{i}."
        wqkv, bqkv = np.asarray(P[s + "Attention_0/Dense_0/kernel"]), np.asarray(P[s + "Attention_0/Dense_0/bias"])
        for j, nme in enumerate(("q_proj", "k_proj", "v_proj")):  # transformers 5.x ViT parameter names
            sd[o + f"attention.{nme}.weight"] = t(wqkv[:, j * D:(j + 1) * D].T)
            sd[o + f"attention.{nme}.bias"] = t(bqkv[j * D:(j + 1) * D])
        sd[o + "attention.o_proj.weight"] = t(np.asarray(P[s + "Attention_0/Dense_1/kernel"]).T)
        sd[o + "attention.o_proj.bias"] = t(P[s + "Attention_0/Dense_1/bias"])
        sd[o + "layernorm_before.weight"] = t(P[s + "LayerNorm_0/scale"])
        sd[o + "layernorm_before.bias"] = t(P[s + "LayerNorm_0/bias"])
        sd[o + "layernorm_after.weight"] = t(P[s + "LayerNorm_1/scale"])
        sd[o + "layernorm_after.bias"] = t(P[s + "LayerNorm_1/bias"])
        sd[o + "mlp.fc1.weight"] = t(np.asarray(P[s + "TransformerMLP_0/fc1/kernel"]).T)
        sd[o + "mlp.fc1.bias"] = t(P[s + "TransformerMLP_0/fc1/bias"])
        sd[o + "mlp.fc2.weight"] = t(np.asarray(P[s + "TransformerMLP_0/fc2/kernel"]).T)
        sd[o + "mlp.fc2.bias"] = t(P[s + "TransformerMLP_0/fc2/bias"])
    sd["layernorm.weight"] = t(P["encoder/LayerNorm_0/scale"])
    sd["layernorm.bias"] = t(P["encoder/LayerNorm_0/bias"])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m


def hf_forward(m, images):
    with torch.no_grad():
        return m(pixel_values=torch.from_numpy(np.asarray(images, np.float64).transpose(0, 3, 1, 2))).last_hidden_state.numpy()
