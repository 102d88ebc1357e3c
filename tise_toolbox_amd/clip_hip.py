"""The CLIP ViT-B/32 towers on the hand-written kernels of csrc/clip_ops.hip.

``HipTowers(model)`` takes a ``clip_model.CLIP`` (real OpenAI parameters or the seeded stand-ins), keeps fp16 copies
of its parameters in the layouts the kernels read, and offers ``encode_image`` / ``encode_text`` with the signatures
of the module's own methods.  Every matrix product -- the patch embedding (as a GEMM over ``tise_patchify_f16``'s
patch matrix), the q/k/v, output, MLP and final projections -- is ``tise_gemm_f16`` with bias / QuickGELU / residual
fused in its epilogue; LayerNorm, attention and the token assembly are the other kernels of that file.  PyTorch only
allocates the buffers.  Arithmetic: fp16 tensors, fp32 accumulation and statistics, i.e. what the fp16 model
``clip.load`` serves on a GPU computes (third-party `clip`, model.py: ``convert_weights`` + fp32 LayerNorm).
Parity of the towers themselves stays UNPINNED (no `clip` package, weights or vocabulary here); the tests compare
against ``clip_model.CLIP`` run in fp32 through PyTorch-ROCm.
"""
import ctypes

import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm(a, w, bias=None, residual=None, act=0, out=None):
    """out = act(a @ w.T + bias) + residual;  a (M, K), w (N, K) fp16 CUDA, row strides arbitrary multiples of 8."""
    assert a.dtype == w.dtype == torch.float16 and a.is_cuda and a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1]
    assert a.stride(1) == 1 and w.stride(1) == 1
    m, k = a.shape
    n = w.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float16, device=a.device)
    _lib.call("tise_gemm_f16", _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(residual),
              residual.stride(0) if residual is not None else 0, _p(out), out.stride(0), m, n, k, int(act), _stream())
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    assert x.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.empty_like(x)
    _lib.call("tise_layernorm_f16", _p(x), x.stride(0), _p(gamma), _p(beta), _p(out), out.stride(0), x.shape[0], x.shape[1],
              ctypes.c_float(eps), _stream())
    return out


def attention(qkv, batch, seq, heads, causal):
    """qkv (batch*seq, 3*heads*64) fp16 contiguous -> (batch*seq, heads*64)."""
    assert qkv.dtype == torch.float16 and qkv.is_contiguous() and qkv.shape == (batch * seq, 3 * heads * 64)
    out = torch.empty((batch * seq, heads * 64), dtype=torch.float16, device=qkv.device)
    _lib.call("tise_attention_f16", _p(qkv), batch, seq, heads, 64, 1 if causal else 0, _p(out), _stream())
    return out


class _Block:
    def __init__(self, blk, dev):
        h = lambda t: t.detach().to(dev, torch.float16).contiguous()
        self.ln1 = (h(blk.ln_1.weight), h(blk.ln_1.bias), blk.ln_1.eps)
        self.ln2 = (h(blk.ln_2.weight), h(blk.ln_2.bias), blk.ln_2.eps)
        self.w_in, self.b_in = h(blk.attn.in_proj_weight), h(blk.attn.in_proj_bias)
        self.w_out, self.b_out = h(blk.attn.out_proj.weight), h(blk.attn.out_proj.bias)
        self.w_fc, self.b_fc = h(blk.mlp.c_fc.weight), h(blk.mlp.c_fc.bias)
        self.w_pr, self.b_pr = h(blk.mlp.c_proj.weight), h(blk.mlp.c_proj.bias)
        self.heads = blk.attn.heads

    def __call__(self, x, batch, seq, causal):
        """x (batch*seq, width) -> x + attn(ln_1(x)); then + mlp(ln_2(.))   (clip model.py ResidualAttentionBlock)."""
        y = layernorm(x, *self.ln1)
        qkv = gemm(y, self.w_in, self.b_in)
        o = attention(qkv, batch, seq, self.heads, causal)
        x = gemm(o, self.w_out, self.b_out, residual=x)
        y = layernorm(x, *self.ln2)
        f = gemm(y, self.w_fc, self.b_fc, act=1)                    # QuickGELU in the epilogue
        return gemm(f, self.w_pr, self.b_pr, residual=x)


class HipTowers:
    def __init__(self, model, device=None):
        if not torch.cuda.is_available():
            raise _lib.TiseLibraryError("HipTowers needs an MI355X: there is no CPU path")
        _lib.load()
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.device = dev
        h = lambda t: t.detach().to(dev, torch.float16).contiguous()
        v = model.visual
        self.patch = v.conv1.kernel_size[0]
        self.width_v = v.conv1.out_channels
        self.w_patch = h(v.conv1.weight.flatten(1))                 # (width, 3 * P * P): columns (c, ky, kx)
        self.cls, self.pos_v = h(v.class_embedding), h(v.positional_embedding)
        self.ln_pre = (h(v.ln_pre.weight), h(v.ln_pre.bias), v.ln_pre.eps)
        self.ln_post = (h(v.ln_post.weight), h(v.ln_post.bias), v.ln_post.eps)
        self.proj_v = h(v.proj.t())                                  # (out_dim, width): features @ proj
        self.blocks_v = [_Block(b, dev) for b in v.transformer.resblocks]
        self.table = h(model.token_embedding.weight)
        self.pos_t = h(model.positional_embedding)
        self.ln_final = (h(model.ln_final.weight), h(model.ln_final.bias), model.ln_final.eps)
        self.proj_t = h(model.text_projection.t())
        self.blocks_t = [_Block(b, dev) for b in model.transformer.resblocks]
        self.width_t = self.table.shape[1]
        self.logit_scale = model.logit_scale

    @torch.no_grad()
    def encode_image(self, image):
        """(B, 3, 224, 224) preprocessed images -> (B, 512) fp16 (clip model.py VisionTransformer.forward)."""
        x = image.to(self.device, torch.float16).contiguous()
        b, _, r, _ = x.shape
        g = r // self.patch
        npatch = g * g
        pm = torch.empty((b * npatch, 3 * self.patch * self.patch), dtype=torch.float16, device=self.device)
        _lib.call("tise_patchify_f16", _p(x), b, r, self.patch, _p(pm), _stream())
        pe = gemm(pm, self.w_patch)                                  # conv1 (no bias)
        seq = npatch + 1
        tok = torch.empty((b * seq, self.width_v), dtype=torch.float16, device=self.device)
        _lib.call("tise_vit_tokens_f16", _p(pe), _p(self.cls), _p(self.pos_v), b, npatch, self.width_v, _p(tok), _stream())
        xs = layernorm(tok, *self.ln_pre)
        for blk in self.blocks_v:
            xs = blk(xs, b, seq, causal=False)
        idx = torch.arange(b, device=self.device, dtype=torch.int64) * seq          # the class token of every image
        c = torch.empty((b, self.width_v), dtype=torch.float16, device=self.device)
        _lib.call("tise_gather_rows_f16", _p(xs), _p(idx), b, self.width_v, _p(c), _stream())
        return gemm(layernorm(c, *self.ln_post), self.proj_v)

    @torch.no_grad()
    def encode_text(self, text):
        """(B, 77) int token ids -> (B, 512) fp16 (clip model.py CLIP.encode_text: features at the end-of-text token =
        the position of the largest id)."""
        text = text.to(self.device)
        b, seq = text.shape
        tok32 = text.to(torch.int32).contiguous()
        x = torch.empty((b * seq, self.width_t), dtype=torch.float16, device=self.device)
        _lib.call("tise_text_tokens_f16", _p(tok32), _p(self.table), _p(self.pos_t), b * seq, seq, self.width_t, _p(x), _stream())
        for blk in self.blocks_t:
            x = blk(x, b, seq, causal=True)
        idx = (torch.arange(b, device=self.device, dtype=torch.int64) * seq + text.argmax(-1).to(torch.int64)).contiguous()
        e = torch.empty((b, self.width_t), dtype=torch.float16, device=self.device)
        _lib.call("tise_gather_rows_f16", _p(x), _p(idx), b, self.width_t, _p(e), _stream())
        return gemm(layernorm(e, *self.ln_final), self.proj_t)

    def parameters(self):                                            # so that callers can ask for the dtype
        yield self.table
