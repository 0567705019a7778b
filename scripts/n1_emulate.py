"""Row N1, numpy emulation (CPU, no GPU): which binary16 roundings of the frozen M3AE encoder cost the policy logits their accuracy?
Each rounding point of the f16 pipeline (patch operands, ln_1 / ln_2 outputs, the four weight matrices, q|k|v, softmax numerators, attention output, hidden
activation) is switched on alone and in groups; the policy behind it is the fp64 oracle.  Output of the round-5 runs: profiles/r5_n1_emulation.txt.
usage: python scripts/n1_emulate.py 0,1,2 [config,config,...]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from arp_amd import m3ae, synth_policy as S
from arp_amd.train import PolicyConfig
from oracle import arpdt_torch as O, m3ae_np as M

def q16(v): return np.asarray(v, np.float64).astype(np.float16).astype(np.float64)
def hilo(v):
    v = np.asarray(v, np.float64); hi = q16(v); return hi + q16(v - hi)

ALL = ["patch", "ln1", "w_qkv", "qkv_out", "p", "ao", "w_out", "ln2", "w_fc1", "hid", "w_fc2"]

def enc(P, cfg, images, on, layers_on=None):
    """on: set of rounding points applied (binary16); others exact"""
    r = lambda name, v, li=None: q16(v) if (name in on and (layers_on is None or li is None or li in layers_on)) else v
    g = lambda k: np.asarray(P[k], np.float64)
    D, hd = cfg.width, cfg.width // cfg.heads
    x = M.patchify(np.asarray(images, np.float64), cfg.patch)
    n, L, _ = x.shape
    x = r("patch", x) @ r("patch", g("image_embedding/kernel")) + g("image_embedding/bias") + M.sincos_2d(D, L) + g("encoder_image_type_embedding")[0]
    x = np.concatenate([np.broadcast_to(g("cls_token"), (n, 1, D)), x], axis=1)
    T = L + 1
    for i in range(cfg.layers):
        p = f"encoder/Block_{i}/"
        y = r("ln1", M._ln(x, g(p + "LayerNorm_0/scale"), g(p + "LayerNorm_0/bias")), i)
        qkv = r("qkv_out", y @ r("w_qkv", g(p + "Attention_0/Dense_0/kernel"), i) + g(p + "Attention_0/Dense_0/bias"), i).reshape(n, T, 3, cfg.heads, hd)
        qq, k, v = (qkv[:, :, j].transpose(0, 2, 1, 3) for j in range(3))
        s = qq @ k.transpose(0, 1, 3, 2) * hd ** -0.5
        s = np.exp(s - s.max(-1, keepdims=True))
        l = s.sum(-1, keepdims=True)
        s = r("p", s, i)              # the kernel rounds the un-normalised exp to binary16 for the PV MFMA, divides by the f32 row sum afterwards
        y = ((s @ v) / l).transpose(0, 2, 1, 3).reshape(n, T, D)
        x = x + r("ao", y, i) @ r("w_out", g(p + "Attention_0/Dense_1/kernel"), i) + g(p + "Attention_0/Dense_1/bias")
        y = r("ln2", M._ln(x, g(p + "LayerNorm_1/scale"), g(p + "LayerNorm_1/bias")), i)
        y = r("hid", M._gelu_tanh(y @ r("w_fc1", g(p + "TransformerMLP_0/fc1/kernel"), i) + g(p + "TransformerMLP_0/fc1/bias")), i)
        x = x + y @ r("w_fc2", g(p + "TransformerMLP_0/fc2/kernel"), i) + g(p + "TransformerMLP_0/fc2/bias")
    return M._ln(x, g("encoder/LayerNorm_0/scale"), g("encoder/LayerNorm_0/bias"))

ecfg, eocfg = m3ae.EncoderConfig(), M.EncConfig()
pcfg, pocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
B, T = 2, pcfg.window
seeds = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
configs = {"all": set(ALL)}
for nme in ALL: configs["only_" + nme] = {nme}
configs["acts"] = {"patch", "ln1", "qkv_out", "p", "ao", "ln2", "hid"}
configs["weights"] = {"w_qkv", "w_out", "w_fc1", "w_fc2"}
configs["attn_side"] = {"ln1", "w_qkv", "qkv_out", "p", "ao", "w_out"}
configs["mlp_side"] = {"ln2", "w_fc1", "hid", "w_fc2"}
if len(sys.argv) > 2: configs = {k: v for k, v in configs.items() if k in sys.argv[2].split(",")}
res = {k: [] for k in configs}
for seed in seeds:
    EP = S.m3ae_params(eocfg, seed=50 + seed); Pp = S.policy_params(pcfg, seed=60 + seed)
    rng = np.random.default_rng(70 + seed)
    frames = S.normalized_frames(B * T, 256, seed=80 + seed).reshape(B, T, 256, 256, 3)
    act = rng.integers(0, pcfg.n_actions, (B, T)).astype(np.int32); rtg = rng.random((B, T, 1)).astype(np.float32)
    Pt = {k: torch.from_numpy(v).double() for k, v in Pp.items()}
    def logits(codes):
        o = O.forward(Pt, pocfg, torch.from_numpy(np.asarray(codes, np.float64).reshape(B, T, ecfg.tokens, ecfg.width)), torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
        return o["action_pred"].numpy(), o["return_pred"].numpy()
    ref_codes = enc(EP, eocfg, frames.reshape(-1, 256, 256, 3), set())
    ra, rr = logits(ref_codes)
    for name, on in configs.items():
        t0 = time.time()
        c = enc(EP, eocfg, frames.reshape(-1, 256, 256, 3), on)
        a, r_ = logits(c)
        e = max(np.abs(a - ra).max(), np.abs(r_ - rr).max())
        ce = np.abs(c - ref_codes)
        res[name].append(e)
        print(f"seed {seed} {name:14s} logits err {e:.3e}  codes max {ce.max():.2e} rms {np.sqrt((ce**2).mean()):.2e}  ({time.time()-t0:.0f}s)", flush=True)
for k, v in res.items():
    print(f"{k:14s} max {max(v):.3e} mean {np.mean(v):.3e} rms {np.sqrt(np.mean(np.square(v))):.3e}")
