"""Labelling pass, numpy emulation (CPU): what a binary16 RESIDUAL STREAM would cost the rewards (bytes are what the pass is short of).
The f16 pipeline as the kernels round it (operands of every GEMM, q|k|v, softmax numerators, attention output, hidden activation), with the residual stream
kept in f32 (the product) or rounded to binary16 after every residual add.  Result (profiles/r5_resid16_emulation.txt): 3.8e-5 -> 9.5e-5 max cosine error on
16 frames -- at north_star's 1e-4, so the f32 residual stream stays.
usage: python scripts/resid16_emulate.py [ViT-B/32] [frames]"""
import sys, numpy as np
sys.path.insert(0, ".")
from arp_amd import clip, synth
from oracle import clip_np as C
from oracle import preprocess as pp

def q(v): return np.asarray(v, np.float64).astype(np.float16).astype(np.float64)
def q32(v): return np.asarray(v, np.float64).astype(np.float32).astype(np.float64)

def enc(W, cfg, x, resid16):
    n = x.shape[0]; P, G, D = cfg.patch, cfg.grid, cfg.width
    rq = q if resid16 else q32
    p = q(x.reshape(n, 3, G, P, G, P).transpose(0, 2, 4, 1, 3, 5).reshape(n, G * G, 3 * P * P))
    x = p @ q(W["visual.conv1.weight"].reshape(D, -1)).T
    cls = np.broadcast_to(W["visual.class_embedding"], (n, 1, D))
    x = np.concatenate([cls, x], axis=1) + W["visual.positional_embedding"]
    x = rq(C.layer_norm(x, W["visual.ln_pre.weight"], W["visual.ln_pre.bias"]))
    H = cfg.heads; hd = D // H
    for i in range(cfg.layers):
        g = lambda k: W[f"visual.transformer.resblocks.{i}." + k]
        h = q(C.layer_norm(x, g("ln_1.weight"), g("ln_1.bias")))
        qkv = q(h @ q(g("attn.in_proj_weight")).T + g("attn.in_proj_bias"))
        qq, kk, vv = np.split(qkv, 3, axis=-1)
        sh = lambda a: a.reshape(n, -1, H, hd).transpose(0, 2, 1, 3)
        qq, kk, vv = sh(qq), sh(kk), sh(vv)
        s = (qq @ kk.transpose(0, 1, 3, 2)) * hd ** -0.5
        pr = q(C._softmax(s))
        o = q((pr @ vv).transpose(0, 2, 1, 3).reshape(n, -1, D))
        x = rq(x + o @ q(g("attn.out_proj.weight")).T + g("attn.out_proj.bias"))
        h = q(C.layer_norm(x, g("ln_2.weight"), g("ln_2.bias")))
        h = q(C.quick_gelu(h @ q(g("mlp.c_fc.weight")).T + g("mlp.c_fc.bias")))
        x = rq(x + h @ q(g("mlp.c_proj.weight")).T + g("mlp.c_proj.bias"))
    c = C.layer_norm(x[:, 0], W["visual.ln_post.weight"], W["visual.ln_post.bias"])
    return q(c) @ q(W["visual.proj"])

name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
heavy = len(sys.argv) > 3
cfg = clip.MODELS[name]; ocfg = C.ClipConfig(patch=cfg.patch)
Wt = synth.clip_weights(ocfg, seed=0)
if heavy:
    rng = np.random.default_rng(5)
    for k in list(Wt):
        if k.startswith("visual.transformer") and k.endswith("weight") and Wt[k].ndim == 2:
            Wt[k] = (Wt[k] * rng.standard_t(3, Wt[k].shape) / 1.7).astype(np.float32)
W = C.cast_weights(Wt, np.float64)
fr = np.concatenate([synth.procgen_like_frames(n - n // 2, seed=1), synth.noise_frames(n // 2, 256, 256, seed=7)])
tok = synth.prompt_tokens(1, 8, seed=2)
x = pp.preprocess(fr).astype(np.float64)
t = C.encode_text(W, ocfg, tok)
ref = C.rewards_from_features(W, C.encode_image(W, ocfg, x), t) / 100
for r16 in (False, True):
    got = C.rewards_from_features(W, enc(W, ocfg, x, r16), t) / 100
    e = np.abs(got - ref)
    print(name, "resid f16" if r16 else "resid f32", f"max {e.max():.3e} rms {np.sqrt((e**2).mean()):.3e}", flush=True)
