"""`VADSpectralV2` / `VADMobileNetV2` with the reference's interfaces, computed by HIP kernels.

Drop-in surface kept from the reference (train/models.py):
  * constructor arguments of `VADSpectralV2` (:1262-1275) and `VADMobileNetV2` (:1096-1113);
  * `forward(waveforms (B,1,1,L)) -> (B,1,1,L)` (:745-756), `training_step` / `validation_step`
    (:681-685, :766-804) returning a 0-d loss that backpropagates, the four logged metric names;
  * `.stft`, `.model.encoder/.decoder/.bottleneck_attention/.vad`, `.H_padding/.W_padding`, `hparams`;
  * the 159 `state_dict` keys and shapes, so reference checkpoints load unchanged.

The nn.Conv2d / BatchNorm2d / Linear / LayerNorm objects in the tree are parameter containers only
(they also reproduce the reference's default initialisation order); their `forward` is never used:
all arithmetic goes through `noise_amd.engine` into libnoise_hip.so.  Parameters live in one flat
fp32 buffer (the `nn.Parameter`s are views), so AdamW and the data-parallel all-reduce are one
kernel / one collective each.
"""
import copy
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import ptr
from .engine import DenoiserPlan, LOSS_MAX_BLOCKS
from .stft import STFT

try:  # Lightning is optional: the class is a LightningModule when it is installed
    import lightning.pytorch as pl
    _LightningModule = pl.LightningModule
except Exception:  # pragma: no cover - depends on the environment
    class _LightningModule(nn.Module):
        """The small part of LightningModule's surface this model uses."""

        def __init__(self, *args, **kwargs):
            super().__init__()
            self._hparams = _AttrDict()
            self.logged_metrics = {}

        def save_hyperparameters(self, **kw):
            import inspect
            frame = inspect.currentframe().f_back
            info = inspect.getargvalues(frame)
            self._hparams = _AttrDict({n: copy.deepcopy(info.locals[n]) for n in info.args if n != "self"})

        @property
        def hparams(self):
            return self._hparams

        def log_dict(self, d, **kw):
            self.logged_metrics = dict(d)

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **overrides):
            """`LightningModule.load_from_checkpoint` as train/export.py:8 uses it: a Lightning checkpoint is a dict with
            "hyper_parameters" (the constructor arguments `save_hyperparameters` recorded) and "state_dict".  The file is
            read with `weights_only=True` (tensors and plain containers only, nothing in it is executed)."""
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=True)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(overrides)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"], strict=strict)
            return model

        def save_checkpoint(self, checkpoint_path, **extra):
            """Writes the two entries of a Lightning checkpoint that `load_from_checkpoint` reads."""
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            torch.save({"hyper_parameters": dict(self.hparams), "state_dict": sd, **extra}, checkpoint_path)


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _make_divisible(v, divisor=8, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


# ----------------------------------------------------------------------------- parameter containers
def _conv_norm_act(cin, cout, k, stride, groups, norm, act, transposed=False):
    """[conv, BatchNorm2d?, activation?] with torchvision's ConvNormActivation child numbering."""
    pad = tuple((kk - 1) // 2 for kk in k) if isinstance(k, (list, tuple)) else (k - 1) // 2
    cls = nn.ConvTranspose2d if transposed else nn.Conv2d
    layers = [cls(cin, cout, tuple(k) if isinstance(k, (list, tuple)) else k, tuple(stride), pad, groups=groups,
                  bias=not norm)]
    if norm:
        layers.append(nn.BatchNorm2d(cout))
    if act:
        layers.append(nn.ReLU6(inplace=True))
    seq = nn.Sequential(*layers)
    seq.out_channels = cout
    return seq


class InvertedResidual(nn.Module):
    """Parameter tree of the reference's InvertedResidual (train/models.py:348-407)."""

    def __init__(self, inp, oup, kernel_size, stride, expand_ratio, transposed=False):
        super().__init__()
        self.stride = stride
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride[-1] == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(_conv_norm_act(inp, hidden, 1, (1, 1), 1, True, True))
        layers.append(_conv_norm_act(hidden, hidden, kernel_size, stride, hidden, True, True, transposed))
        layers.append(nn.Conv2d(hidden, oup, 1, 1, 0, bias=False))
        layers.append(nn.BatchNorm2d(oup))
        self.conv = nn.Sequential(*layers)
        self.out_channels = oup


class _SelfAttention(nn.Module):
    def __init__(self, n_head, d_model, d_k, d_v):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(d_model, n_head * d_v, bias=False)
        self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)


class _FeedForward(nn.Module):
    def __init__(self, d_in, d_hid):
        super().__init__()
        self.w_1 = nn.Linear(d_in, d_hid)
        self.w_2 = nn.Linear(d_hid, d_in)
        self.layer_norm = nn.LayerNorm(d_in, eps=1e-6)


class _EncoderLayer(nn.Module):
    def __init__(self, d_model, d_inner, n_head, d_k, d_v):
        super().__init__()
        self.slf_attn = _SelfAttention(n_head, d_model, d_k, d_v)
        self.pos_ffn = _FeedForward(d_model, d_inner)


class TransformerEncoder(nn.Module):
    """Parameter tree of the reference's TransformerEncoder with n_position=0 (train/models.py:172-225)."""

    def __init__(self, d_model, d_inner, n_layers, n_head, d_k, d_v):
        super().__init__()
        self.layer_stack = nn.ModuleList([_EncoderLayer(d_model, d_inner, n_head, d_k, d_v) for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.d_model = d_model


# ----------------------------------------------------------------------------- the U-Net
class VADMobileNetV2(_LightningModule):
    SUPPORTED_SAMPLING_RATE = 16000

    def __init__(self, nhead: int, num_layers: int, dropout: float, width_mult: float = 1.0,
                 layer_config: Optional[List] = None, round_nearest: int = 8, block=None, norm_layer=None,
                 pad_value: float = 0, mr_stft_lambda: float = 0.5, fft_sizes: List[int] = [512, 1024, 2048],
                 hop_lengths: List[int] = [50, 120, 240], win_lengths: List[int] = [240, 600, 1200]):
        super().__init__()
        self.save_hyperparameters()
        if block is not None or norm_layer is not None:
            raise NotImplementedError("custom block / norm_layer are not supported by the kernel schedule")
        if dropout != 0.0:
            raise NotImplementedError("dropout != 0 on the residual branches is not scheduled (reference config uses 0)")
        self.pad_value = pad_value
        self.H_padding = 0
        self.W_padding = 0
        self.searching = False
        self.found = False
        if layer_config is None:
            layer_config = [[1, 32, 1, [3, 3], [2, 2]], [1, 16, 1, [3, 3], [1, 1]], [6, 24, 2, [3, 3], [2, 2]],
                            [6, 32, 3, [3, 3], [2, 2]], [6, 64, 4, [3, 3], [2, 2]], [6, 96, 3, [3, 3], [1, 1]],
                            [6, 160, 3, [3, 3], [2, 2]], [6, 320, 1, [3, 3], [1, 1]]]
        # the reference pops the first row off the caller's list; keep a private copy instead
        cfg = copy.deepcopy(list(layer_config))
        if len(cfg) == 0 or len(cfg[0]) != 5:
            raise ValueError(f"layer_config should be non-empty or a 4-element list, got {cfg}")
        self._first_row = cfg[0]
        self.layer_config = cfg[1:]
        _, c0, _, k, s = self._first_row
        kernels = [list(k)] + [list(r[3]) for r in self.layer_config]
        if any(kk not in ([3, 3], [1, 3]) for kk in kernels):
            raise NotImplementedError("only 3x3 and 1x3 kernels are scheduled")
        # The tuned per-layer engine (training, graphs, the block-fused bf16 path) serves the spectral configuration
        # family: 3x3 kernels, one block per row without residual connection, one encoder layer.  Anything else in the
        # reference's constructor range (the waveform-domain train/configs/mobilenetv2.yaml: 1x3 kernels, repeated
        # residual blocks, two encoder layers, d_model 128) is served by the shape-generic forward of `generic.py`.
        why = []
        if any(kk != [3, 3] for kk in kernels):
            why.append("non-3x3 kernels")
        if num_layers != 1:
            why.append(f"num_layers = {num_layers}")
        c0 = _make_divisible(c0 * width_mult, round_nearest)
        self._enc_plan = [dict(kind="first", cin=1, cout=c0, stride=tuple(s))]
        self._dec_plan = [dict(kind="last", cin=c0, cout=1, stride=tuple(s))]
        self.encoder = nn.ModuleList([_conv_norm_act(1, c0, k, s, 1, True, True)])
        self.decoder = nn.ModuleList([_conv_norm_act(c0, 1, k, s, 1, False, False, transposed=True)])
        cin = c0
        for t, c, n, k, s in self.layer_config:
            cout = _make_divisible(c * width_mult, round_nearest)
            for i in range(n):
                stride = tuple(s) if i == 0 else (1, 1)
                self.encoder.append(InvertedResidual(cin, cout, k, stride, t, transposed=False))
                self.decoder.insert(0, InvertedResidual(cout, cin, k, stride, t, transposed=True))
                self._enc_plan.append(dict(kind="ir", cin=cin, cout=cout, t=t, stride=stride, hidden=int(round(cin * t))))
                self._dec_plan.insert(0, dict(kind="irT", cin=cout, cout=cin, t=t, stride=stride,
                                              hidden=int(round(cout * t))))
                if stride[-1] == 1 and cin == cout:
                    why.append("residual inverted-residual blocks")
                cin = cout
        self.untuned = ", ".join(dict.fromkeys(why)) or None
        self.attn_dropout = 0.1        # ScaledDotProductAttention's own default (train/models.py:22), train mode only
        self.attn_keep_masks = None    # optional explicit keep masks, one (B, heads, n, n) uint8 tensor per encoder layer
        self.grad_sync = None          # generic training step: f(list of gradients) averaging them over ranks (dist.make_grad_mean)
        hidden = self.layer_config[-1][1]
        self.n_head = nhead
        self.d_k = hidden // nhead
        self.bottleneck_attention = TransformerEncoder(hidden, hidden, num_layers, nhead, self.d_k, self.d_k)
        self.vad = nn.Linear(hidden, 1)
        self._plans = {}
        # Counts the updates of parameters / BatchNorm buffers made through raw device pointers (fused optimizer step,
        # running statistics of a training forward, state_dict loads): such writes do not bump a tensor's `_version`,
        # and the block-fused bf16 path keys its folded weight images on this counter as well.
        self._weights_epoch = 0

    def weights_changed(self):
        self._weights_epoch += 1

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.weights_changed()
        return out

    def block_plan(self):
        return self._enc_plan, self._dec_plan

    # ---- padding search (closed form of train/models.py:628-639; found once, like the reference)
    def _out_size(self, n, axis):
        strides = [b["stride"][axis] for b in self._enc_plan]
        for s in strides:
            n = (n - 1) // s + 1
        for s in reversed(strides):
            n = (n - 1) * s + 1
        return n

    def _search_padding(self, H, W):
        self.searching = True
        while self._out_size(H + self.H_padding, 0) < H:
            self.H_padding += 1
        while self._out_size(W + self.W_padding, 1) < W:
            self.W_padding += 1
        self.found = True

    def reset_padding(self):
        """Forget the searched padding (the reference keeps the first one for the model's lifetime)."""
        if any(q.pinned for q in self._plans.values()):
            raise RuntimeError("reset_padding: a HIP graph captured with the current padding is still alive "
                               "(drop the function returned by graphed_forward first)")
        self.H_padding = self.W_padding = 0
        self.searching = self.found = False
        self._plans.clear()

    def plan(self, B, H, W, device, fused=False):
        """Buffers + schedule for one geometry: the per-layer fp32 engine, or (fused=True) the block-fused bf16
        inference kernels."""
        if self.untuned:
            raise NotImplementedError(f"the tuned kernel schedule does not cover this configuration ({self.untuned}); "
                                      "only the shape-generic forward (`forward` / `_forward`) serves it")
        if not self.searching and not self.found:
            self._search_padding(H, W)
        key = (B, H, W, self.H_padding, self.W_padding, str(device), bool(fused))
        p = self._plans.get(key)
        if p is None:
            # plans hold activations: keep only a few geometries alive.  A plan captured in a HIP graph is pinned and
            # never evicted (the graph has its buffer addresses baked in).
            loose = [k for k, q in self._plans.items() if not q.pinned]
            if len(loose) >= 4:
                self._plans.pop(loose[0])
            if fused:
                from .fused import FusedDenoiser
                p = FusedDenoiser(self, B, H, W, device)
            else:
                p = DenoiserPlan(self, B, H, W, device)
            self._plans[key] = p
        return p

    def _forward(self, x):
        """x (B,1,H,W), already normalised -> filters (B,1,H,W), vad (B,1)   (train/models.py:583-612).
        Inference-only entry point kept for interface parity; training goes through VADSpectralV2."""
        _lib.require_cuda(x)
        if self.untuned:
            from .generic import GenericForward
            return GenericForward(self).forward(x)
        B, _, H, W = x.shape
        p = self.plan(B, H, W, x.device)
        xin = x.reshape(B, H, W).contiguous().float()
        one = torch.full((B,), 1.0 - 1e-6, device=x.device)      # kernel divides by (absmax + 1e-6)
        p.forward(xin, one, self.training)
        from . import ops
        last = p.dec[-1][0]
        filt = torch.empty(B, H, W, device=x.device)
        ops.last_convT_fwd(last.src_a, last.src_b, last.conv.weight, last.conv.bias, None, None, None, filt, B, last.Hi,
                           last.Wi, last.Cin, last.stride, H, W)
        return filt.view(B, 1, H, W), p.vad.view(B, 1).clone()

    @torch.no_grad()
    def forward(self, waveforms):
        """Base.forward (train/models.py:576-581): normalise by the per-clip maximum, run the U-Net, gate by
        sigmoid(1e4 * vad), undo the normalisation.  (B,1,H,W) -> (B,1,H,W)."""
        _lib.require_cuda(waveforms)
        from .generic import normalize, finish
        B, _, H, W = waveforms.shape
        xn, mx = normalize(waveforms)
        filt, vad = self._forward(xn)
        return finish(filt.reshape(B, filt.shape[-2], filt.shape[-1]), None, vad, mx, H, W)

    SEGMENT = 8000  # `_shared_step` reshapes every batch to 0.5 s segments (train/models.py:643-644)

    @torch.no_grad()
    def _shared_step(self, batch, batch_idx, stage):
        """Base._shared_step (train/models.py:641-679) as an evaluation: U-Net forward on 8000-sample segments, then
        smooth-L1 + lambda * multi-resolution STFT loss + VAD cross-entropy (`noise_amd.losses`), logged under the
        reference's four names.  The returned loss carries no graph: this path has no backward kernels."""
        from .generic import normalize, finish
        from .losses import waveform_loss
        noisy, clean, _ = batch
        _lib.require_cuda(noisy, clean)
        noisy = noisy.reshape(-1, 1, 1, self.SEGMENT)
        clean = clean.reshape(-1, 1, 1, self.SEGMENT).contiguous().float()
        B = noisy.shape[0]
        xn, mx = normalize(noisy)
        filt, vad = self._forward(xn)
        enhanced = finish(filt.reshape(B, 1, self.SEGMENT), None, None, mx, 1, self.SEGMENT)     # _denormalize, no gate
        hp = self.hparams
        loss, l1, mr, vl = waveform_loss(enhanced, clean, vad, hp.mr_stft_lambda, tuple(hp.fft_sizes), tuple(hp.hop_lengths),
                                         tuple(hp.win_lengths))
        self.log_dict({f"{stage}_loss": loss, f"{stage}_waveform_loss": l1, f"{stage}_spectral_loss": mr,
                       f"{stage}_vad_loss": vl}, on_step=False, on_epoch=True, prog_bar=True, logger=True, sync_dist=True)
        return loss

    def validation_step(self, batch, batch_idx):
        return self._shared_step(batch, batch_idx, "val")

    def training_step(self, batch, batch_idx):
        """Base.training_step (train/models.py:681-682): the loss of `_shared_step` in train mode; `loss.backward()` fills
        every parameter's `.grad` (the backward pass has already run on the shape-generic kernels)."""
        from .generic_train import StepFn
        noisy, clean, _ = batch
        _lib.require_cuda(noisy, clean)
        if not self.training:
            raise RuntimeError("training_step in eval mode: the backward pass implements train-mode BatchNorm only")
        params = list(self.parameters())
        loss, all4 = StepFn.apply(self, noisy, clean, *params)
        self.log_dict({"train_loss": loss, "train_waveform_loss": all4[1], "train_spectral_loss": all4[2],
                       "train_vad_loss": all4[3]}, on_step=False, on_epoch=True, prog_bar=True, logger=True, sync_dist=True)
        return loss


# ----------------------------------------------------------------------------- autograd bridge
class _StepFn(torch.autograd.Function):
    """loss = shared_step(noisy, clean); the native backward runs inside forward (the training step
    always backpropagates) and its flat gradient is handed to autograd in backward."""

    @staticmethod
    def forward(ctx, model, noisy, clean, keep, *params):
        losses = model._native_step(noisy, clean, keep, want_grad=True)
        ctx.model = model
        # the native backward has already run into the shared work buffer: keep this step's gradients with this
        # step's graph node, so a later training_step cannot replace them before .backward() is called (305 KB)
        ctx.flat_grads = model._work_grads.clone()
        all4 = losses.clone()
        ctx.mark_non_differentiable(all4)
        return losses[0].clone(), all4

    @staticmethod
    def backward(ctx, g_loss, g_all):
        model = ctx.model
        flat = ctx.flat_grads * g_loss
        return (None, None, None, None) + tuple(flat[o:o + n].view(s) for (o, n, s) in model._param_slices)


class VADSpectralV2(_LightningModule):
    SUPPORTED_SAMPLING_RATE = 16000
    SEGMENT = 8000  # the training step reshapes every batch to 0.5 s segments (train/models.py:768-769)

    def __init__(self, n_fft: int, hop_length: Optional[int], win_length: Optional[int], window: str, nhead: int,
                 num_layers: int, dropout: float, layer_config: Optional[List], pad_value: float, max_frames: int):
        super().__init__()
        self.save_hyperparameters()
        hop_length = hop_length if hop_length is not None else n_fft // 4
        win_length = win_length if win_length is not None else n_fft
        self.stft = STFT(max_frames=max_frames, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window)
        self.model = VADMobileNetV2(nhead=nhead, num_layers=num_layers, dropout=dropout, layer_config=layer_config,
                                    pad_value=pad_value)
        if self.model.untuned:
            raise NotImplementedError(f"the spectral training engine does not cover this configuration "
                                      f"({self.model.untuned})")
        self._flat_params = None
        self._flat_grads = None
        self._work_grads = None
        self._param_slices = []
        self._step_bufs = {}
        # "bf16": eval-mode forwards run the block-fused kernels (bf16 storage, fp32 accumulation, BatchNorm folded);
        # training and "fp32" run the per-layer fp32 kernels
        self.inference_dtype = "fp32"
        self.stft_fast_math = True     # see forward()
        # True: the training step transforms the network input with the dense kernel that follows the reference's conv1d
        # accumulation order (see STFT.forward): the per-clip normaliser then equals the reference's and the gradients
        # match it to ~1e-6 of each tensor.  False takes the FFT kernel for this input too (0.29 ms per step less at
        # 1280 segments); results differ by fp32 rounding of the spectrum, which the normaliser (a max over
        # near-floor bins) magnifies to ~1e-2 of a gradient tensor in the worst case measured.
        self.reference_order_stft = True
        self.attn_keep_mask = None     # optional explicit (B,heads,n,n) uint8 keep-mask for the next training step
        self.attn_seed = 0x5EED

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.model.weights_changed()
        return out

    # ------------------------------------------------------------------ flat parameter storage
    def _trainable(self):
        return [p for p in self.model.parameters()]

    def _ensure_flat(self):
        params = self._trainable()
        dev = params[0].device
        ok = self._flat_params is not None and self._flat_params.device == dev
        if ok:
            base = self._flat_params.data_ptr()
            for p, (o, n, s) in zip(params, self._param_slices):
                if p.data_ptr() != base + 4 * o:
                    ok = False
                    break
        if not ok:
            total = sum(p.numel() for p in params)
            flat = torch.empty(total, device=dev, dtype=torch.float32)
            self._param_slices = []
            o = 0
            for p in params:
                n = p.numel()
                flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = flat[o:o + n].view(p.shape)
                self._param_slices.append((o, n, tuple(p.shape)))
                o += n
            self._flat_params = flat
            self._flat_grads = torch.zeros(total, device=dev, dtype=torch.float32)
            self._work_grads = torch.zeros(total, device=dev, dtype=torch.float32)
            self._grad_index = {id(p): i for i, p in enumerate(params)}
        return params

    def _grad_views(self, flat):
        idx, sl = self._grad_index, self._param_slices

        def grad_of(p):
            o, n, s = sl[idx[id(p)]]
            return flat[o:o + n].view(s)
        return grad_of

    def bind_grads(self):
        """Point every parameter's .grad at its slice of the flat gradient buffer (fused-step mode)."""
        params = self._ensure_flat()
        g = self._grad_views(self._flat_grads)
        for p in params:
            p.grad = g(p)
        return self._flat_params, self._flat_grads

    # ------------------------------------------------------------------ inference
    def _spectral(self, log_magnitudes, absmax, training, keep=None, seed=0):
        B, _, H, W = log_magnitudes.shape
        if self._fused_eval(training):
            plan = self.model.plan(B, H, W, log_magnitudes.device, fused=True)
            enh, vad = plan.forward(log_magnitudes.view(B, H, W), absmax)
            return plan, enh, vad
        plan = self.model.plan(B, H, W, log_magnitudes.device)
        enh, vad = plan.forward(log_magnitudes.view(B, H, W), absmax, training, keep, seed)
        return plan, enh, vad

    def _fused_eval(self, training):
        if self.inference_dtype not in ("fp32", "bf16"):
            raise ValueError(f"inference_dtype must be 'fp32' or 'bf16', got {self.inference_dtype!r}")
        return self.inference_dtype == "bf16" and not training

    def _forward(self, log_magnitudes):
        """(B,1,F,T) -> enhanced log-magnitudes (B,1,F,T), vad (B,1)   (train/models.py:758-764)."""
        _lib.require_cuda(log_magnitudes)
        self._ensure_flat()
        lm = log_magnitudes.contiguous().float()
        B = lm.shape[0]
        absmax = torch.empty(B, device=lm.device, dtype=torch.float32)
        _lib.call("noise_absmax", ptr(lm), ptr(absmax), B, lm[0].numel(), _lib.stream())
        _, enh, vad = self._spectral(lm, absmax, self.training)
        return enh.view_as(lm).clone(), vad.view(B, 1).clone()

    @torch.no_grad()
    def forward(self, waveforms):
        """(B,1,1,L) -> enhanced (B,1,1,L') with L' = hop*(L//hop)   (train/models.py:745-756)."""
        _lib.require_cuda(waveforms)
        self._ensure_flat()
        # Inference takes the hardware-rate log / exp / sine / cosine and the polynomial angle of the FFT kernels in both
        # precisions: measured against float64 the spectrum is within 1.3e-6 of its largest bin (library functions:
        # 2.3e-7; the parity bar is 1e-5) and the reconstruction within 1e-7, 10-25 % faster.  `stft_fast_math = False`
        # restores the library functions; the training step always uses them.
        fast = self._fused_eval(self.training) or self.stft_fast_math
        lm, ph, absmax = self.stft.forward(waveforms, return_absmax=True, fast_math=fast)
        _, enh, vad = self._spectral(lm, absmax, self.training, seed=self._next_seed())
        B, _, H, W = lm.shape
        return self.stft._inverse_raw(enh.view(B, 1, H, W), ph, vad, fast_math=fast)

    def graphed_forward(self, example):
        """Capture the eval forward for inputs shaped like `example` in a HIP graph and return `run(waveforms)`.
        A single 0.5 s window is ~60 launches of a few microseconds each, so the streaming case is launch-bound;
        replaying the captured graph removes the per-launch host cost.  The returned tensor is a static buffer
        that the next call overwrites."""
        if self.training:
            raise RuntimeError("graphed_forward captures the eval forward: call .eval() first")
        _lib.require_cuda(example)
        static_in = example.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):       # padding search, plan allocation, code-object loads happen here
                self.forward(static_in)
        cur.wait_stream(side)
        # The graph bakes in raw device addresses: the plan's activation buffers, the flat parameter buffer and the
        # kernel-side layouts of the STFT bases.  The closure owns references to all of them (so the caching allocator
        # cannot hand their memory to anyone else while `run` is alive), the plan is pinned against eviction, and `run`
        # refuses to replay once the model has moved its parameters or bases elsewhere (`.to()`, a new basis).
        B, _, _, L = static_in.shape
        fused = self._fused_eval(False)
        plan = self.model.plan(B, self.stft.cutoff, self.stft.num_frames(L), static_in.device, fused=fused)
        plan.pinned += 1
        flat, derived = self._flat_params, self.stft._derived
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            static_out = self.forward(static_in)
        # the bf16 path bakes its folded half-precision weight images (and their BatchNorm statistics) into the graph
        wkey = plan.weights_key() if fused else None
        if self.model.plan(B, self.stft.cutoff, self.stft.num_frames(L), static_in.device, fused=fused) is not plan or \
                self._flat_params is not flat or self.stft._derived is not derived:
            plan.pinned -= 1
            raise RuntimeError("graphed_forward: buffers moved during capture")
        model = self

        class _Replay:
            """Callable owner of the captured graph; dropping it unpins the plan."""

            def __call__(self, waveforms):
                if model._flat_params is not flat or model.stft._derived is not derived or \
                        flat.data_ptr() != model._trainable()[0].data_ptr():
                    raise RuntimeError("graphed_forward: the model's parameters or STFT bases were reallocated after "
                                       "the capture (.to(), load into new storage): capture again")
                if wkey is not None and plan.weights_key() != wkey:
                    raise RuntimeError("graphed_forward: parameters or BatchNorm statistics changed after the capture "
                                       "(optimizer step, training forward, load_state_dict); the bf16 graph holds "
                                       "folded copies of the old values: capture again")
                static_in.copy_(waveforms)
                graph.replay()
                return static_out

            def __del__(self):
                plan.pinned -= 1

            keep_alive = (plan, flat, derived, static_in)

        return _Replay()

    # ------------------------------------------------------------------ training
    def _next_seed(self):
        self.attn_seed = (self.attn_seed * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return self.attn_seed

    def _native_step(self, noisy, clean, keep, want_grad):
        """One pass of the training/validation step on 0.5 s segments.  Returns the device tensor
        [loss, vad_loss, waveform_loss, spectral_loss]; with want_grad the parameter gradients are left in
        `_work_grads` (autograd mode) or `_flat_grads` (fused mode, see `train_step_fused`)."""
        _lib.require_cuda(noisy, clean)
        self._ensure_flat()
        if want_grad and not self.training:
            # the backward kernels implement train-mode BatchNorm (batch statistics); gradients of an eval-mode
            # forward (running statistics) would be silently wrong
            raise RuntimeError("gradients are only available in train mode: call .train() (frozen-BatchNorm "
                               "fine-tuning is not scheduled)")
        seg = self.SEGMENT
        noisy = noisy.reshape(-1, seg).contiguous().float()
        clean = clean.reshape(-1, seg).contiguous().float()
        B = noisy.shape[0]
        dev = noisy.device
        st = _lib.stream()
        stft = self.stft
        # the network input keeps the reference's accumulation order (see STFT.forward): its per-clip normaliser is
        # sensitive to fp32 rounding, and gradients are compared against the reference's
        lm_n, ph, absmax = stft.forward(noisy.view(B, 1, 1, seg), return_absmax=True, reference_order=self.reference_order_stft,
                                        fast_math=False)
        lm_c, _ = stft.forward(clean.view(B, 1, 1, seg), want_phase=False, fast_math=False)
        _, _, H, W = lm_n.shape
        training = self.training
        if keep is not None:
            n_tok = self.model.plan(B, H, W, dev).n_tok
            want = (B, self.model.n_head, n_tok, n_tok)
            if tuple(keep.shape) != want or keep.dtype != torch.uint8 or keep.device != dev or not keep.is_contiguous():
                raise ValueError(f"attn_keep_mask must be a contiguous uint8 tensor of shape {want} on {dev}, got "
                                 f"{tuple(keep.shape)} {keep.dtype} on {keep.device}")
        plan, enh, vad = self._spectral(lm_n, absmax, training, keep, self._next_seed())
        wave = stft._inverse_raw(enh.view(B, 1, H, W), ph, None, fast_math=False)
        Lw = wave.shape[-1]
        if Lw != seg:
            raise RuntimeError(f"iSTFT output length {Lw} != segment length {seg}: hop_length must divide {seg} "
                               "(the reference's loss fails the same way)")
        key = (B, H, W, str(dev))
        bufs = self._step_bufs.get(key)
        if bufs is None:
            self._step_bufs.clear()
            bufs = dict(label=torch.empty(B, device=dev), d_enh=torch.empty(B, H, W, device=dev),
                        d_wave=torch.empty(B, seg, device=dev), d_vad=torch.empty(B, device=dev),
                        spec_partial=torch.empty(LOSS_MAX_BLOCKS, device=dev, dtype=torch.float64),
                        wave_partial=torch.empty((seg + 1) // 2, device=dev, dtype=torch.float64),
                        out4=torch.empty(4, device=dev))
            self._step_bufs[key] = bufs
        import ctypes
        n_spec, n_wave = ctypes.c_int(0), ctypes.c_int(0)
        g = want_grad
        _lib.call("noise_vad_label", ptr(clean), ptr(bufs["label"]), B, seg, 160, st, nbytes=4 * B * seg)
        _lib.call("noise_spectral_l1", ptr(enh), ptr(lm_c), ptr(bufs["d_enh"]) if g else None, ptr(bufs["spec_partial"]),
                  ctypes.byref(n_spec), B * H * W, st, nbytes=4 * B * H * W * (3 if g else 2))
        _lib.call("noise_wave_pair_l1", ptr(wave), ptr(clean), ptr(bufs["d_wave"]) if g else None,
                  ptr(bufs["wave_partial"]), ctypes.byref(n_wave), B, seg, st, nbytes=4 * B * seg * (3 if g else 2))
        _lib.call("noise_loss_finalize", ptr(bufs["spec_partial"]), n_spec.value, B * H * W, ptr(bufs["wave_partial"]),
                  n_wave.value, B, seg, ptr(vad), ptr(bufs["label"]), ptr(bufs["d_vad"]) if g else None, ptr(bufs["out4"]),
                  st)
        if want_grad:
            stft._inverse_bwd(bufs["d_wave"], wave, enh.view(B, 1, H, W), ph, d_lm=bufs["d_enh"], accumulate=True)
            plan.backward(bufs["d_enh"], bufs["d_vad"], self._grad_views(self._grad_target))
        return bufs["out4"]

    def _take_keep_mask(self):
        """The explicit dropout mask, if one was set, serves exactly one TRAINING step; eval / validation steps run
        without dropout and leave it in place."""
        if not self.training:
            return None
        keep, self.attn_keep_mask = self.attn_keep_mask, None
        return keep

    def _shared_step(self, batch, batch_idx, stage):
        noisy, clean, _ = batch
        params = self._ensure_flat()
        keep = self._take_keep_mask()
        # eval-mode steps (validation outside no_grad, frozen statistics) are evaluated without a graph: the native
        # backward implements train-mode BatchNorm only
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in params):
            self._grad_target = self._work_grads
            loss, all4 = _StepFn.apply(self, noisy, clean, keep, *params)
        else:
            all4 = self._native_step(noisy, clean, keep, want_grad=False).clone()
            loss = all4[0]
        self.log_dict({f"{stage}_loss": loss, f"{stage}_vad_loss": all4[1], f"{stage}_waveform_loss": all4[2],
                       f"{stage}_spectral_loss": all4[3]}, on_step=False, on_epoch=True, prog_bar=True, logger=True,
                      sync_dist=True)
        return loss

    def training_step(self, batch, batch_idx):
        return self._shared_step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        return self._shared_step(batch, batch_idx, "val")

    def train_step_fused(self, noisy, clean, optimizer, all_reduce=None):
        """Forward + loss + backward (+ gradient all-reduce) + AdamW without autograd.  Returns the device
        tensor [loss, vad_loss, waveform_loss, spectral_loss] of this rank's batch (no host sync)."""
        self._ensure_flat()
        self._grad_target = self._flat_grads
        out4 = self._native_step(noisy, clean, self._take_keep_mask(), want_grad=True)
        scale = 1.0
        if all_reduce is not None:
            scale = all_reduce(self._flat_grads)
        optimizer.step(grad_scale=scale)
        return out4

    def configure_optimizers(self):
        """AdamW with the reference configuration (train/configs/spectralv2.yaml:99-111)."""
        from .optim import FusedAdamW
        return FusedAdamW(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
