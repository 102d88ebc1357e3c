"""GPU: the hand-written CLIP tower kernels (csrc/clip_ops.hip) against PyTorch fp32 on the same fp16 inputs, and the
towers built from them (tise_toolbox_amd/clip_hip.py) against clip_model.CLIP run in fp32 (the towers themselves:
parity unpinned -- no `clip` package / weights here; this pins the KERNELS to the published architecture's math)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n,k,bias,res,act", [(12800, 768, 768, True, True, 0), (300, 2304, 768, True, False, 0),
                                                (1000, 3072, 768, True, False, 1), (257, 512, 3072, False, False, 0),
                                                (5, 64, 64, True, True, 1), (2560, 136, 128, True, True, 0),
                                                (77 * 33, 2048, 512, True, False, 1),
                                                # >= 768 tiles of 256 x 256: the large-tile kernel, with M and N tails
                                                (8192 - 37, 6144 - 8, 128, True, True, 1), (157696, 1536, 512, True, False, 0)])
def test_gemm_f16_matches_fp32_matmul(cuda_device, m, n, k, bias, res, act):
    from tise_toolbox_amd import clip_hip
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    a = (torch.randn((m, k), generator=g) * 0.7).half().to(cuda_device)
    w = (torch.randn((n, k), generator=g) * k ** -0.5).half().to(cuda_device)
    b = (torch.randn(n, generator=g) * 0.3).half().to(cuda_device) if bias else None
    r = torch.randn((m, n), generator=g).half().to(cuda_device) if res else None
    out = clip_hip.gemm(a, w, b, r, act)
    ref = a.float() @ w.float().t()
    if bias:
        ref = ref + b.float()
    if act:
        ref = ref * torch.sigmoid(1.702 * ref)
    ref_h = ref.half().float()                                    # the epilogue rounds to fp16 BEFORE the residual add
    if res:
        ref_h = (ref_h + r.float())
    err = (out.float() - ref_h).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref_h.abs().max().item()), err       # one fp16 rounding of the result
    again = clip_hip.gemm(a, w, b, r, act)
    assert torch.equal(out, again)
    # strided operands (a slice of a wider matrix) and a preallocated output
    wide = torch.zeros((m, k + 64), dtype=torch.float16, device=cuda_device)
    wide[:, 64:] = a
    out2 = torch.empty((m, n + 8), dtype=torch.float16, device=cuda_device)
    clip_hip.gemm(wide[:, 64:], w, b, r, act, out=out2[:, :n])
    assert torch.equal(out2[:, :n], out)


@pytest.mark.parametrize("rows,c", [(1000, 768), (77, 512), (3, 1024), (130, 64)])
def test_layernorm_f16(cuda_device, rows, c):
    from tise_toolbox_amd import clip_hip
    g = torch.Generator(device="cpu").manual_seed(rows + c)
    x = (torch.randn((rows, c), generator=g) * 3 + 1).half().to(cuda_device)
    ga = (1 + 0.2 * torch.randn(c, generator=g)).half().to(cuda_device)
    be = (0.3 * torch.randn(c, generator=g)).half().to(cuda_device)
    got = clip_hip.layernorm(x, ga, be, 1e-5)
    ref = torch.nn.functional.layer_norm(x.float(), (c,), ga.float(), be.float(), 1e-5)
    assert (got.float() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("batch,seq,heads,causal", [(7, 50, 12, False), (5, 77, 8, True), (1, 1, 2, False), (3, 80, 1, True)])
def test_attention_f16(cuda_device, batch, seq, heads, causal):
    from tise_toolbox_amd import clip_hip
    g = torch.Generator(device="cpu").manual_seed(seq + heads)
    e = heads * 64
    qkv = torch.randn((batch * seq, 3 * e), generator=g).half().to(cuda_device)
    got = clip_hip.attention(qkv, batch, seq, heads, causal)
    q, k, v = qkv.float().view(batch, seq, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal).transpose(1, 2).reshape(batch * seq, e)
    assert (got.float() - ref).abs().max().item() <= 3e-3 * max(1.0, ref.abs().max().item())


def test_towers_match_fp32_module(cuda_device):
    """HipTowers (fp16 kernels) vs clip_model.CLIP in fp32 on PyTorch-ROCm, same (stand-in) parameters rounded to fp16:
    cosine similarity of every embedding >= 0.9995 and the image-text similarity matrix within 2e-3 -- the level of the
    fp16 model itself (the library-kernel fp16 module is measured alongside)."""
    from tise_toolbox_amd import clip_hip, clip_model
    model = clip_model.build_clip()
    model_h = model.to(cuda_device).half()
    ref = clip_model.build_clip().to(cuda_device)
    ref.load_state_dict({k: v.float() for k, v in model_h.state_dict().items()})       # fp16-rounded parameters, fp32 math
    towers = clip_hip.HipTowers(model_h)
    g = torch.Generator(device="cpu").manual_seed(0)
    img = torch.randn((19, 3, 224, 224), generator=g).to(cuda_device)
    tok = clip_model.HashTokenizer()([f"a photo of thing number {i} on a table" for i in range(23)]).to(cuda_device)
    with torch.no_grad():
        fi, ft = towers.encode_image(img), towers.encode_text(tok)
        ri, rt = ref.encode_image(img.half().float()), ref.encode_text(tok)
        li, lt = model_h.encode_image(img.half()), model_h.encode_text(tok)
    assert fi.shape == (19, 512) and ft.shape == (23, 512) and fi.dtype == torch.float16

    def cos(a, b):
        return torch.nn.functional.cosine_similarity(a.float(), b.float(), dim=-1)
    print("hip vs fp32: image cos min", cos(fi, ri).min().item(), "text", cos(ft, rt).min().item(),
          "| library fp16 vs fp32:", cos(li, ri).min().item(), cos(lt, rt).min().item())
    assert cos(fi, ri).min().item() >= 0.9995 and cos(ft, rt).min().item() >= 0.9995
    n = lambda t: t.float() / t.float().norm(dim=-1, keepdim=True)
    sim, sim_ref = n(fi) @ n(ft).t(), n(ri) @ n(rt).t()
    assert (sim - sim_ref).abs().max().item() <= 2e-3
    assert torch.equal(towers.encode_image(img), fi)                # repeatable


def test_towers_are_batch_invariant_and_cli_switch(cuda_device, monkeypatch):
    """An embedding must not depend on what else is in the batch (every GEMM row and every (sequence, head) is
    computed in a fixed order), so RP / PA scores do not depend on --batch-size; TISE_CLIP=torch selects the module."""
    from tise_toolbox_amd import RP_coco, clip_hip, clip_model
    towers, scale = RP_coco.build_towers(None, cuda_device)
    assert isinstance(towers, clip_hip.HipTowers) and abs(scale - 1 / 0.07) < 1e-3
    g = torch.Generator(device="cpu").manual_seed(1)
    img = torch.randn((9, 3, 224, 224), generator=g).to(cuda_device).half()
    tok = clip_model.HashTokenizer()([f"caption {i} of a red bus" for i in range(11)]).to(cuda_device)
    fi, ft = towers.encode_image(img), towers.encode_text(tok)
    one_i = torch.cat([towers.encode_image(img[i:i + 1]) for i in range(9)])
    one_t = torch.cat([towers.encode_text(tok[i:i + 1]) for i in range(11)])
    assert torch.equal(one_i, fi) and torch.equal(one_t, ft)
    monkeypatch.setenv("TISE_CLIP", "torch")
    mod, _ = RP_coco.build_towers(None, cuda_device)
    assert isinstance(mod, clip_model.CLIP)
    cos = torch.nn.functional.cosine_similarity(mod.encode_image(img).float(), fi.float(), dim=-1)
    assert cos.min().item() >= 0.9995


def test_text_context_is_truncated_at_the_longest_caption_exactly(cuda_device, monkeypatch):
    """Round 4: the text transformer is causal and CLIP.encode_text reads the feature at the end-of-text token, so the
    padding behind it cannot reach the result: RP_coco.embed_texts sorts a chunk's captions by length and encodes every
    batch at its longest caption instead of 77 tokens.  Row for row the arithmetic is that of the 77-token run (per-token
    LayerNorm and GEMM rows, the same keys per query, masked keys contribute exact zeros): BIT-IDENTICAL embeddings as
    long as both runs use the same GEMM kernel (launches of < 768 large tiles: the batches here), in order, for captions
    of 1 .. 60 words (the tokenizer clips at 75), duplicates and the empty caption."""
    from tise_toolbox_amd import RP_coco, clip_hip, clip_model
    towers, _ = RP_coco.build_towers(None, cuda_device)
    assert isinstance(towers, clip_hip.HipTowers)
    rng = np.random.default_rng(3)
    words = [f"w{k}" for k in range(500)]
    caps = [" ".join(rng.choice(words, int(n))) for n in rng.integers(1, 61, size=301)] + ["", "a", "a"]
    tok = clip_model.HashTokenizer()
    seen = []
    orig = towers.encode_text

    def spy(t):
        seen.append(tuple(t.shape))
        return orig(t)
    monkeypatch.setattr(towers, "encode_text", spy)
    short = RP_coco.embed_texts(towers, tok, caps, cuda_device, 32)
    assert max(s[1] for s in seen) <= 62 and min(s[1] for s in seen) <= 24 and sum(s[0] for s in seen) == len(caps)
    seen.clear()
    monkeypatch.setenv("TISE_CLIP_TRUNCATE", "0")
    full = RP_coco.embed_texts(towers, tok, caps, cuda_device, 32)
    assert all(s[1] == clip_model.CONTEXT_LENGTH for s in seen)
    assert short.shape == full.shape == (len(caps), 512) and torch.equal(short, full)
    assert torch.equal(short[-1], short[-2])                          # duplicates: the same embedding wherever they land
    # a large batch (the 256 x 256 GEMM kernel on one side only): equal to fp16 rounding
    monkeypatch.delenv("TISE_CLIP_TRUNCATE")
    many = caps * 8
    a = RP_coco.embed_texts(towers, tok, many, cuda_device, 2048)
    cos = torch.nn.functional.cosine_similarity(a.float(), full.repeat(8, 1).float(), dim=-1)
    assert cos.min().item() >= 0.99999


def test_clip_preprocess_on_the_device_equals_pillow_preprocess(cuda_device):
    """clip._transform on the device (round 5, RP_coco.embed_paths): Pillow-exact BICUBIC 8-bit resample (tise_resize_u8,
    filter 1) + CenterCrop + ToTensor / Normalize through a table against clip_model.preprocess (Pillow on the host, the
    reference's ``preprocess(Image.open(p))``): the resized uint8 image bit for bit against Pillow and the oracle, the fp32
    network input EQUAL element for element -- square 256 x 256 (the metric's images), the identity size, an up-scale and
    two non-square shapes (crop)."""
    import numpy as np
    from PIL import Image
    from oracle import resize_oracle
    from tests import _cases
    from tise_toolbox_amd import clip_model, device
    dev = cuda_device
    rng = np.random.default_rng(7)
    for (h, w) in ((256, 256), (224, 224), (128, 128), (300, 200), (180, 333)):
        imgs = np.stack([_cases.smooth_images(1, h, w, seed=int(rng.integers(1 << 20)))[0] if k % 2 == 0 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
                         for k in range(5)])
        nh, nw, top, left = clip_model.preprocess_geometry(h, w)
        x, u8 = device.resize_u8_lut(torch.from_numpy(imgs).to(dev), (nh, nw), clip_model.preprocess_lut(), filter="bicubic", return_u8=True)
        for k in range(len(imgs)):
            want_u8 = np.asarray(Image.fromarray(imgs[k]).resize((nw, nh), Image.BICUBIC))
            assert np.array_equal(u8[k].cpu().numpy(), want_u8), (h, w, k)
            assert np.array_equal(want_u8, resize_oracle.resize_u8(imgs[k], nh, nw, "bicubic"))
        got = clip_model.preprocess_device(torch.from_numpy(imgs).to(dev)).cpu()
        want = torch.stack([clip_model.preprocess(Image.fromarray(im)) for im in imgs])
        assert got.shape == (5, 3, 224, 224) and torch.equal(got, want), (h, w, float((got - want).abs().max()))


def test_rp_ring_feed_equals_dataloader_feed(cuda_device, tmp_path):
    """RP_coco.embed_paths: the ring feed + device preprocess against the DataLoader feed (Pillow preprocess on worker
    processes): the same embeddings bit for bit; an RGBA file follows RP_coco.py:64's convert("RGB") on both roads, and under
    PA's rule (no conversion before the resize, PA.py:34) sends the directory to the DataLoader road."""
    import numpy as np
    from PIL import Image
    from tests import _cases
    from tise_toolbox_amd import RP_coco
    dev = cuda_device
    model, _ = RP_coco.build_towers(None, dev)
    imgs = _cases.smooth_images(37, 256, 256, seed=4)
    paths = []
    for i, im in enumerate(imgs):
        p = tmp_path / f"{i}.png"
        (Image.fromarray(im).convert("RGBA") if i == 11 else Image.fromarray(im)).save(p)
        paths.append(str(p))
    a = RP_coco.embed_paths(model, paths, dev, 16, workers=3, feed="ring")
    b = RP_coco.embed_paths(model, paths, dev, 16, workers=2, feed="dataloader")
    assert a.shape == (37, 512) and torch.equal(a, b)
    c = RP_coco.embed_paths(model, paths, dev, 16, workers=3, feed="ring", convert_first=False)      # PA's rule: RGBA -> DataLoader road
    d = RP_coco.embed_paths(model, paths, dev, 16, workers=2, feed="dataloader", convert_first=False)
    assert torch.equal(c, d)
