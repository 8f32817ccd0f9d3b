"""fp32 CPU restatement of the Oobleck VAE decoder (latent -> waveform) and encoder (waveform -> latent) (test oracle).

PARITY UNPINNED.  The arithmetic lives in third-party ``diffusers``
(``diffusers.models.AutoencoderOobleck``; unpinned at
/root/reference/pyproject.toml:25; absent from /root/reference and from this image).
This file restates the published architecture from the reference's own in-tree MLX
restatement:
  * acestep/models/mlx/vae_model.py:24-55   Snake1d
  * acestep/models/mlx/vae_model.py:62-87   OobleckResidualUnit
  * acestep/models/mlx/vae_model.py:119-142 OobleckDecoderBlock
  * acestep/models/mlx/vae_model.py:190-230 OobleckDecoder
  * acestep/models/mlx/vae_model.py:92-116, 148-187, 285-310  OobleckEncoderBlock / OobleckEncoder / encode_and_sample
    (same unpinned status)
  * acestep/models/mlx/vae_convert.py:18-34 weight_norm fusion (w = g * v / (||v|| + 1e-9))
in PyTorch NCL layout with ``F.conv1d`` / ``F.conv_transpose1d``.  Weight names are
the keys of ``AutoencoderOobleck.state_dict()`` as consumed by vae_convert.py:62-127
(``decoder.block.{i}.res_unit{j}.conv1.weight_g`` ...).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class VaeConfig:
    """Decoder-side fields of AutoencoderOobleck.config (vae_model.py:251-264, from_pytorch_config :322-336).

    ``downsampling_ratios`` is run-time data from ``checkpoints/vae/config.json``; ACE-Step's
    hop is 1920 (handler/conditioning_target.py:47,53) so the synthetic default is
    [2,4,4,6,10] (decoder order [10,6,4,4,2]) rather than the MLX class default (hop 2048).
    """

    decoder_channels: int = 128
    decoder_input_channels: int = 64
    encoder_hidden_size: int = 128
    audio_channels: int = 2
    channel_multiples: Tuple[int, ...] = (1, 2, 4, 8, 16)
    downsampling_ratios: Tuple[int, ...] = (2, 4, 4, 6, 10)

    @property
    def upsampling_ratios(self) -> Tuple[int, ...]:
        return tuple(self.downsampling_ratios[::-1])

    @property
    def hop(self) -> int:
        return int(math.prod(self.downsampling_ratios))

    def block_dims(self) -> List[Tuple[int, int, int]]:
        """(in_ch, out_ch, stride) per decoder block, vae_model.py:213-221."""
        cm = [1] + list(self.channel_multiples)
        s = self.upsampling_ratios
        n = len(s)
        return [(self.decoder_channels * cm[n - i], self.decoder_channels * cm[n - i - 1], s[i]) for i in range(n)]


    def encoder_block_dims(self) -> List[Tuple[int, int, int]]:
        """(in_ch, out_ch, stride) per encoder block, vae_model.py:171-179."""
        cm = [1] + list(self.channel_multiples)
        return [(self.encoder_hidden_size * cm[i], self.encoder_hidden_size * cm[i + 1], s) for i, s in enumerate(self.downsampling_ratios)]


def fuse_weight_norm(g: Tensor, v: Tensor, eps: float = 1e-9) -> Tensor:
    """vae_convert.py:18-34: norm over all dims but 0, reshaped like g."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return g * v / (norm + eps)


def snake(x: Tensor, alpha: Tensor, beta: Tensor) -> Tensor:
    """vae_model.py:52-55 (logscale): x + 1/(exp(beta)+1e-9) * sin(exp(alpha)*x)^2; params [1,C,1]."""
    a = torch.exp(alpha)
    b = torch.exp(beta)
    return x + (b + 1e-9).reciprocal() * torch.sin(a * x).pow(2)


def _w(w: Dict[str, Tensor], base: str) -> Tensor:
    if base + ".weight" in w:  # already fused
        return w[base + ".weight"]
    return fuse_weight_norm(w[base + ".weight_g"], w[base + ".weight_v"])


def _ident(x: Tensor) -> Tensor:
    return x


def _bf16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def residual_unit(w: Dict[str, Tensor], p: str, x: Tensor, dilation: int, q=_ident, qs=None, q_out=None) -> Tensor:
    """vae_model.py:62-87: x + conv_k1(snake2(conv_k7_dil(snake1(x)))), pad = 3*dilation.

    Rounding points of the bf16-storage emulation: ``q`` after every stored tensor; ``qs`` (default ``q``) on the k = 7 result, whose
    only reader is snake2; ``q_out`` (default ``q``) on the unit's output.  The HIP path keeps both of the latter in fp32 when the
    producer's epilogue applies the reader's Snake (``qs = q_out = identity``, emulate_bf16="native").
    """
    qs = q if qs is None else qs
    q_out = q if q_out is None else q_out
    y = qs(F.conv1d(q(snake(x, w[p + ".snake1.alpha"], w[p + ".snake1.beta"])), q(_w(w, p + ".conv1")), w[p + ".conv1.bias"],
                    dilation=dilation, padding=3 * dilation))
    y = F.conv1d(q(snake(y, w[p + ".snake2.alpha"], w[p + ".snake2.beta"])), q(_w(w, p + ".conv2")), w[p + ".conv2.bias"])
    return q_out(x + y)


def decoder_block(w: Dict[str, Tensor], p: str, x: Tensor, stride: int, q=_ident, qs=None) -> Tensor:
    """vae_model.py:119-142: snake -> ConvTranspose1d(k=2s, stride=s, pad=ceil(s/2)) -> res units d=1,3,9.

    The block's output has one reader (the next block's Snake, or the output conv's): it leaves through ``qs``."""
    qs = q if qs is None else qs
    x = q(snake(x, w[p + ".snake1.alpha"], w[p + ".snake1.beta"]))
    x = q(F.conv_transpose1d(x, q(_w(w, p + ".conv_t1")), w[p + ".conv_t1.bias"], stride=stride, padding=math.ceil(stride / 2)))
    x = residual_unit(w, p + ".res_unit1", x, 1, q, qs)
    x = residual_unit(w, p + ".res_unit2", x, 3, q, qs)
    x = residual_unit(w, p + ".res_unit3", x, 9, q, qs, qs)
    return x


def _rounders(emulate_bf16):
    """(q, qs): emulate_bf16 False -> no rounding; True -> every stored tensor AND every Snake input rounded to bf16 (the placement of
    a layer-by-layer bf16 model); "native" -> stored tensors rounded, but a tensor whose only reader is a Snake stays fp32 up to that
    Snake and only snake(t) is stored - the rounding points of the HIP path (csrc/vae.hip, ConvArgs::osnake_a)."""
    if not emulate_bf16:
        return _ident, _ident
    if emulate_bf16 == "native":
        return _bf16, _ident
    return _bf16, _bf16


def decode(cfg: VaeConfig, w: Dict[str, Tensor], z: Tensor, emulate_bf16=False) -> Tensor:
    """vae_model.py:190-230: z [B,64,T] -> waveform [B,2,hop*T] (== vae.decode(z).sample).

    ``emulate_bf16=True`` rounds weights and every inter-layer activation to bfloat16 (fp32 arithmetic in
    between): the storage precision of the reference's own GPU VAE (handler/memory_utils.py:157-166, bf16 on
    cuda) and of the HIP path, used to separate kernel error from storage-precision drift.  ``"native"``: see _rounders.
    """
    q, qs = _rounders(emulate_bf16)
    x = qs(F.conv1d(q(z), q(_w(w, "decoder.conv1")), w["decoder.conv1.bias"], padding=3))
    for i, (_cin, _cout, s) in enumerate(cfg.block_dims()):
        x = decoder_block(w, f"decoder.block.{i}", x, s, q, qs)
    x = q(snake(x, w["decoder.snake1.alpha"], w["decoder.snake1.beta"]))
    return F.conv1d(x, q(_w(w, "decoder.conv2")), None, padding=3)


def decoder_weight_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict names/shapes of the decoder half (weight-normed: weight_g / weight_v / bias)."""
    shapes: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cout, cin, k, bias=True):
        shapes[name + ".weight_g"] = (cout, 1, 1)
        shapes[name + ".weight_v"] = (cout, cin, k)
        if bias:
            shapes[name + ".bias"] = (cout,)

    def snk(name, c):
        shapes[name + ".alpha"] = (1, c, 1)
        shapes[name + ".beta"] = (1, c, 1)

    dims = cfg.block_dims()
    conv("decoder.conv1", dims[0][0], cfg.decoder_input_channels, 7)
    for i, (cin, cout, s) in enumerate(dims):
        p = f"decoder.block.{i}"
        snk(p + ".snake1", cin)
        # ConvTranspose1d weight is [in, out, K]; weight_g is [in,1,1] (vae_convert.py:27-30)
        shapes[p + ".conv_t1.weight_g"] = (cin, 1, 1)
        shapes[p + ".conv_t1.weight_v"] = (cin, cout, 2 * s)
        shapes[p + ".conv_t1.bias"] = (cout,)
        for j in (1, 2, 3):
            r = f"{p}.res_unit{j}"
            snk(r + ".snake1", cout)
            conv(r + ".conv1", cout, cout, 7)
            snk(r + ".snake2", cout)
            conv(r + ".conv2", cout, cout, 1)
    snk("decoder.snake1", cfg.decoder_channels)
    conv("decoder.conv2", cfg.audio_channels, cfg.decoder_channels, 7, bias=False)
    return shapes


def encoder_block(w: Dict[str, Tensor], p: str, x: Tensor, stride: int, q=_ident, qs=None, q_out=None) -> Tensor:
    """vae_model.py:92-116: res units d=1,3,9 -> snake -> Conv1d(k=2s, stride=s, pad=ceil(s/2)).  ``qs``: rounding of the tensors read by
    one Snake only (the k = 7 results, the third unit's output); ``q_out``: of the block's output (read by one Snake after the last block)."""
    qs = q if qs is None else qs
    q_out = q if q_out is None else q_out
    x = residual_unit(w, p + ".res_unit1", x, 1, q, qs)
    x = residual_unit(w, p + ".res_unit2", x, 3, q, qs)
    x = residual_unit(w, p + ".res_unit3", x, 9, q, qs, qs)
    x = q(snake(x, w[p + ".snake1.alpha"], w[p + ".snake1.beta"]))
    return q_out(F.conv1d(x, q(_w(w, p + ".conv1")), w[p + ".conv1.bias"], stride=stride, padding=math.ceil(stride / 2)))


def encode_moments(cfg: VaeConfig, w: Dict[str, Tensor], audio: Tensor, emulate_bf16=False) -> Tuple[Tensor, Tensor]:
    """vae_model.py:148-187 + :296-302: audio [B,2,L] -> (mean, std), each [B, 64, L // hop]; std = softplus(scale) + 1e-4."""
    q, qs = _rounders(emulate_bf16)
    x = q(F.conv1d(q(audio), q(_w(w, "encoder.conv1")), w["encoder.conv1.bias"], padding=3))
    dims = cfg.encoder_block_dims()
    for i, (_cin, _cout, s) in enumerate(dims):
        x = encoder_block(w, f"encoder.block.{i}", x, s, q, qs, qs if i + 1 == len(dims) else q)
    x = q(snake(x, w["encoder.snake1.alpha"], w["encoder.snake1.beta"]))
    h = F.conv1d(x, q(_w(w, "encoder.conv2")), w["encoder.conv2.bias"], padding=1)
    mean, scale = h.chunk(2, dim=1)
    return mean, F.softplus(scale) + 1e-4


def encode(cfg: VaeConfig, w: Dict[str, Tensor], audio: Tensor, noise: Tensor = None, emulate_bf16=False) -> Tensor:
    """``vae.encode(audio).latent_dist.sample()`` (handler/vae_encode.py:66): mean + std * noise (noise None -> the mean)."""
    mean, std = encode_moments(cfg, w, audio, emulate_bf16)
    return mean if noise is None else mean + std * noise


def encoder_weight_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict names/shapes of the encoder half (weight-normed: weight_g / weight_v / bias)."""
    shapes: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cout, cin, k):
        shapes[name + ".weight_g"] = (cout, 1, 1)
        shapes[name + ".weight_v"] = (cout, cin, k)
        shapes[name + ".bias"] = (cout,)

    def snk(name, c):
        shapes[name + ".alpha"] = (1, c, 1)
        shapes[name + ".beta"] = (1, c, 1)

    conv("encoder.conv1", cfg.encoder_hidden_size, cfg.audio_channels, 7)
    dims = cfg.encoder_block_dims()
    for i, (cin, cout, s) in enumerate(dims):
        p = f"encoder.block.{i}"
        for j in (1, 2, 3):
            r = f"{p}.res_unit{j}"
            snk(r + ".snake1", cin)
            conv(r + ".conv1", cin, cin, 7)
            snk(r + ".snake2", cin)
            conv(r + ".conv2", cin, cin, 1)
        snk(p + ".snake1", cin)
        conv(p + ".conv1", cout, cin, 2 * s)
    snk("encoder.snake1", dims[-1][1])
    conv("encoder.conv2", cfg.encoder_hidden_size, dims[-1][1], 3)
    return shapes
