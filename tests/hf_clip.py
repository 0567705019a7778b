"""Test helper: build a HuggingFace ``CLIPModel`` (an independent implementation of openai/CLIP)
from an openai-style state dict, to pin the numpy oracle.  Not part of the product."""
import numpy as np
import torch


def build_hf_clip(W, cfg, eos_id):
    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig

    vc = CLIPVisionConfig(hidden_size=cfg.width, intermediate_size=4 * cfg.width, num_hidden_layers=cfg.layers,
                          num_attention_heads=cfg.heads, image_size=cfg.img_res, patch_size=cfg.patch,
                          hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=cfg.embed,
                          attn_implementation="eager")
    tc = CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.txt_width, intermediate_size=4 * cfg.txt_width,
                        num_hidden_layers=cfg.txt_layers, num_attention_heads=cfg.txt_heads,
                        max_position_embeddings=cfg.ctx, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                        projection_dim=cfg.embed, eos_token_id=eos_id, bos_token_id=0, pad_token_id=0,
                        attn_implementation="eager")
    config = CLIPConfig(text_config=tc.to_dict(), vision_config=vc.to_dict(), projection_dim=cfg.embed)
    config._attn_implementation = "eager"
    model = CLIPModel(config).double().eval()
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64).copy())
    sd = {}
    sd["vision_model.embeddings.patch_embedding.weight"] = t(W["visual.conv1.weight"])
    sd["vision_model.embeddings.class_embedding"] = t(W["visual.class_embedding"])
    sd["vision_model.embeddings.position_embedding.weight"] = t(W["visual.positional_embedding"])
    sd["vision_model.pre_layrnorm.weight"] = t(W["visual.ln_pre.weight"])
    sd["vision_model.pre_layrnorm.bias"] = t(W["visual.ln_pre.bias"])
    sd["vision_model.post_layernorm.weight"] = t(W["visual.ln_post.weight"])
    sd["vision_model.post_layernorm.bias"] = t(W["visual.ln_post.bias"])
    sd["visual_projection.weight"] = t(W["visual.proj"].T)
    sd["text_model.embeddings.token_embedding.weight"] = t(W["token_embedding.weight"])
    sd["text_model.embeddings.position_embedding.weight"] = t(W["positional_embedding"])
    sd["text_model.final_layer_norm.weight"] = t(W["ln_final.weight"])
    sd["text_model.final_layer_norm.bias"] = t(W["ln_final.bias"])
    sd["text_projection.weight"] = t(W["text_projection"].T)
    sd["logit_scale"] = t(W["logit_scale"])

    def tower(src, dst, layers, d):
        for i in range(layers):
            s, o = f"{src}resblocks.{i}.", f"{dst}encoder.layers.{i}."
            wq, wk, wv = np.split(W[s + "attn.in_proj_weight"], 3, axis=0)
            bq, bk, bv = np.split(W[s + "attn.in_proj_bias"], 3, axis=0)
            for n_, w_, b_ in (("q", wq, bq), ("k", wk, bk), ("v", wv, bv)):
                sd[o + f"self_attn.{n_}_proj.weight"] = t(w_)
                sd[o + f"self_attn.{n_}_proj.bias"] = t(b_)
            sd[o + "self_attn.out_proj.weight"] = t(W[s + "attn.out_proj.weight"])
            sd[o + "self_attn.out_proj.bias"] = t(W[s + "attn.out_proj.bias"])
            for a, b in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2")):
                sd[o + b + ".weight"] = t(W[s + a + ".weight"])
                sd[o + b + ".bias"] = t(W[s + a + ".bias"])
            sd[o + "mlp.fc1.weight"] = t(W[s + "mlp.c_fc.weight"])
            sd[o + "mlp.fc1.bias"] = t(W[s + "mlp.c_fc.bias"])
            sd[o + "mlp.fc2.weight"] = t(W[s + "mlp.c_proj.weight"])
            sd[o + "mlp.fc2.bias"] = t(W[s + "mlp.c_proj.bias"])

    tower("visual.transformer.", "vision_model.", cfg.layers, cfg.width)
    tower("transformer.", "text_model.", cfg.txt_layers, cfg.txt_width)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    return model


def hf_rewards(model, x_nchw, tokens):
    """logits_per_text[0] from the HF model (== openai ``model(images, text)[1][0]``)."""
    with torch.no_grad():
        out = model(pixel_values=torch.from_numpy(np.asarray(x_nchw, np.float64)),
                    input_ids=torch.from_numpy(np.asarray(tokens, np.int64)),
                    attention_mask=None)
    return (out.logits_per_text[0].numpy(), out.image_embeds.numpy(), out.text_embeds.numpy())
