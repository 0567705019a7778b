"""Generates tests/golden/finetune_tiny.npz (run in the build container; never on the GPU box).

    python tests/golden/make_golden_finetune.py

What pins what: the loss, its two components and the gradient of every trainable tensor come from the REFERENCE CLASS
ITSELF -- /root/reference/finetune_module/clip_multiscale_adapter.py::CLIPMultiscaleAdapter.forward + autograd --
executed here with its missing imports replaced by stand-ins:
  * `clip`      -> `load()` returns a small random-init torch model with openai/CLIP's module layout
                   (visual.transformer.resblocks.N / transformer.resblocks.N, LND activations, encode_image / encode_text);
                   it stands in for the FROZEN towers, whose outputs are recorded and handed to the build as inputs;
  * `kornia`    -> ColorJitter is the identity (the random augmentation is outside the step under test);
  * `torchvision.transforms.functional` -> `normalize` restated ((x - mean) / std), `resize` never called (224x224 input).
None of the stand-ins computes anything the head under test is judged on: everything from the recorded tower features to
the loss and the gradients is the reference's own code.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

L, WV, WT, HID, NA, B, CTX, VOCAB = 2, 64, 64, 64, 15, 5, 77, 100


class _Block(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.ln = nn.LayerNorm(w)
        self.fc = nn.Linear(w, w)

    def forward(self, x):  # [seq, batch, width]
        return x + torch.tanh(self.fc(self.ln(x))) + 0.1 * x.mean(dim=0, keepdim=True)


class _Tower(nn.Module):
    def __init__(self, w, layers):
        super().__init__()
        self.width, self.layers = w, layers
        self.resblocks = nn.Sequential(*[_Block(w) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class _Visual(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, WV, kernel_size=32, stride=32, bias=False)
        self.class_embedding = nn.Parameter(0.1 * torch.randn(WV))
        self.transformer = _Tower(WV, L)
        self.proj = nn.Parameter(WV ** -0.5 * torch.randn(WV, WT))

    def forward(self, x):
        x = self.conv1(x).flatten(2).permute(0, 2, 1)
        x = torch.cat([self.class_embedding.expand(x.shape[0], 1, -1), x], dim=1)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        return x[:, 0, :] @ self.proj


class _StubClip(nn.Module):
    def __init__(self):
        super().__init__()
        self.visual = _Visual()
        self.transformer = _Tower(WT, L)
        self.token_embedding = nn.Embedding(VOCAB, WT)
        self.text_projection = nn.Parameter(WT ** -0.5 * torch.randn(WT, WT))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        x = self.token_embedding(text).permute(1, 0, 2)
        x = self.transformer(x).permute(1, 0, 2)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection


def _install_stubs():
    clip = types.ModuleType("clip")
    clip.load = lambda name, **kw: (_StubClip(), None)
    sys.modules["clip"] = clip
    kornia = types.ModuleType("kornia")
    aug = types.ModuleType("kornia.augmentation")
    aug.ColorJitter = lambda *a, **k: nn.Identity()
    kornia.augmentation = aug
    sys.modules["kornia"], sys.modules["kornia.augmentation"] = kornia, aug
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")

    def normalize(x, mean, std):
        m = torch.tensor(mean, dtype=x.dtype).view(1, 3, 1, 1)
        s = torch.tensor(std, dtype=x.dtype).view(1, 3, 1, 1)
        return (x - m) / s

    def resize(x, size):
        raise AssertionError("resize must not be reached: the fixture feeds 224x224 frames")

    fn.normalize, fn.resize = normalize, resize
    tr.functional = fn
    tv.transforms = tr
    sys.modules["torchvision"], sys.modules["torchvision.transforms"], sys.modules["torchvision.transforms.functional"] = tv, tr, fn


def main(goal=False):
    """goal = True: the goal_conditioned variant (clip_multiscale_adapter.py:208-212,224-230) -> finetune_tiny_goal.npz: image3 takes
    the prompt's place (scores = -||a3 - a_k||, inverse-model input [a1|a3|a2|a3]); the text tower is never run."""
    _install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    from finetune_module.clip_multiscale_adapter import CLIPMultiscaleAdapter
    model = CLIPMultiscaleAdapter(input_dim=WT, hidden_dim=HID, output_dim=WT, action_dim=NA, use_discrete_action=True, use_vip_loss=True,
                                  use_id_loss=True, goal_conditioned=goal)
    for p in model.clip_model.parameters():  # finetune.py:139-140
        p.requires_grad = False
    # move the head off its special init so that every term of the gradient is exercised
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("clip_model."):
                continue
            if n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
        model.image_residual_weight.fill_(0.7)
        model.text_residual_weight.fill_(-0.3)
        model.lambda_id.fill_(1.3)

    rng = np.random.Generator(np.random.PCG64(1))
    frames = [torch.from_numpy(rng.integers(0, 256, (B, 224, 224, 3)).astype(np.float32)) for _ in range(4)]
    tokens = np.zeros((B, 1, CTX), np.int64)
    for b in range(B):
        n = int(rng.integers(2, 8))
        tokens[b, 0, 0] = VOCAB - 2
        tokens[b, 0, 1:1 + n] = rng.integers(1, VOCAB - 2, n)
        tokens[b, 0, 1 + n] = VOCAB - 1  # EOT = largest id -> argmax position
    r = rng.integers(0, 2, (B, 1)).astype(np.int64)
    action = rng.integers(0, NA, (B,)).astype(np.int64)
    batch = {f"image{i}": {"ob": frames[i]} for i in range(4)}
    batch.update({"instruct": torch.from_numpy(tokens), "r": torch.from_numpy(r), "action": torch.from_numpy(action)})

    # record what the frozen towers hand to the head (inputs of the two bias-free Linears, outputs of encode_*)
    rec = {"img_inter": [], "img_final": [], "txt_inter": [], "txt_final": []}
    model.image_intermediate_linear.register_forward_pre_hook(lambda m, a: rec["img_inter"].append(a[0].detach().clone()))
    model.text_intermediate_linear.register_forward_pre_hook(lambda m, a: rec["txt_inter"].append(a[0].detach().clone()))
    enc_i, enc_t = model.clip_model.encode_image, model.clip_model.encode_text

    def wrap_i(x):
        y = enc_i(x)
        rec["img_final"].append(y.detach().float().clone())
        return y

    def wrap_t(x):
        y = enc_t(x)
        rec["txt_final"].append(y.detach().float().clone())
        return y

    model.clip_model.encode_image, model.clip_model.encode_text = wrap_i, wrap_t

    loss = model(batch)
    loss.backward()
    out = {"cfg": np.array([L, WV, WT, WT, HID, NA]), "gamma": model.gamma, "logit_scale": float(model.logit_scale),
           "img_inter": torch.stack(rec["img_inter"]).numpy(), "img_final": torch.stack(rec["img_final"]).numpy(),
           "r": r[:, 0].astype(np.float32), "action": action, "loss": float(loss), "goal_conditioned": int(goal)}
    if not goal:
        out["txt_inter"], out["txt_final"] = rec["txt_inter"][0].numpy(), rec["txt_final"][0].numpy()
    else:  # four image groups (image0..image3), no prompt
        assert len(rec["img_inter"]) == 4 and not rec["txt_inter"]
    for n, p in model.named_parameters():
        if n.startswith("clip_model."):
            assert p.grad is None
            continue
        out["param:" + n] = p.detach().numpy()
        out["grad:" + n] = p.grad.numpy() if p.grad is not None else np.zeros_like(p.detach().numpy())
    # the two components, from the reference's own switches
    with torch.no_grad():
        model.use_id_loss = False
        out["vip_loss"] = float(model(batch))
        model.use_id_loss, model.use_vip_loss = True, False
        out["lambda_id_times_id_loss"] = float(model(batch))
    np.savez_compressed(os.path.join(HERE, "finetune_tiny_goal.npz" if goal else "finetune_tiny.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") and getattr(v, "ndim", 0) else v) for k, v in out.items() if not k.startswith(("param:", "grad:"))})


if __name__ == "__main__":
    main(goal="--goal" in sys.argv)
