"""GPU parity: the HIP path (through the C ABI, via the reference-shaped Python API) against
the fixtures produced by the reference and against the CPU oracle on seeded inputs.

Tolerances (north_star): f32 path — embeddings / dot products within 1e-4, top-k id sets
identical (fp64-adjudicated boundary near-ties reported, not counted), MRR@10 within 1e-4.
bf16 path (the reference's `--fp16` autocast analogue) — looser, stated per test.
"""
import math
import numpy as np
import pytest
import torch

from oracle import encoder_ref, flatip, retrieval_ref
from tests.helpers import NS, items_from_golden, model_from_golden, synth_tokens

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = [("bert_tiny_first", "bert", False), ("bert_tiny_mean_head_norm", "bert", False),
         ("t5_tiny_gtr", "t5", False), ("t5_tiny_gated", "t5", True)]


def build_drmodel(g, arch, gated, dtype="float32"):
    from openmatch.modeling import DRModelForInference, LinearHead
    cfg, lm = model_from_golden(g, arch, gated)
    pooling, has_head, normalize, _ = g["meta"]
    head = None
    if has_head == "1":
        head = LinearHead(128, 128)
        head.linear.weight.data.copy_(torch.from_numpy(g["head_w"]))
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling=str(pooling), head_q=head, head_p=head,
                                normalize=normalize == "1", model_args=NS(encoder_only=arch == "t5", dtype=dtype))
    return model.to(DEV).eval()


@pytest.mark.parametrize("name,arch,gated", CASES)
def test_encoder_f32_matches_reference_fixture(golden, name, arch, gated):
    g = golden(name)
    model = build_drmodel(g, arch, gated)
    for kind in ("p", "q"):
        items = items_from_golden(g, kind, DEV)
        hidden, reps = model.encode_passage(items)
        assert reps.dtype == torch.float32 and reps.is_cuda
        assert np.abs(reps.cpu().numpy() - g[kind + "_reps"]).max() < 1e-4
        if kind == "p":
            m = torch.from_numpy(g["p_attention_mask"][:3]).bool()
            diff = (hidden[:3].cpu() - torch.from_numpy(g["p_hidden"])).abs()
            assert diff[m].max() < 1e-4          # padded positions carry no contract
    out = model(query=items_from_golden(g, "q", DEV), passage=items_from_golden(g, "p", DEV))
    scores = (out.q_reps @ out.p_reps.t()).cpu().numpy()
    assert np.abs(scores - g["scores"]).max() < 1e-4


@pytest.mark.parametrize("name,arch,gated", CASES[:3])
def test_encoder_bf16_close_to_reference_fixture(golden, name, arch, gated):
    """bf16 MFMA path vs the f32 reference: cosine >= 0.999 per embedding (bf16 has 8 mantissa bits)."""
    g = golden(name)
    model = build_drmodel(g, arch, gated, dtype="bfloat16")
    for kind in ("p", "q"):
        _, reps = model.encode_passage(items_from_golden(g, kind, DEV))
        a, b = reps.cpu().double(), torch.from_numpy(g[kind + "_reps"]).double()
        cos = torch.nn.functional.cosine_similarity(a, b, dim=1)
        assert cos.min() > 0.999, cos


@pytest.mark.parametrize("name,arch,gated", [c for c in CASES if c[1] == "bert"][:3])
def test_encoder_f16_close_to_reference_fixture(golden, name, arch, gated):
    """float16 MFMA mode (the reference's `--fp16` is torch.cuda.amp float16) vs the f32 reference: 11 mantissa bits in
    every stored activation -- cosine >= 0.99999 per embedding on the small fixtures (generic GEMM tiles, unfused
    LayerNorm: these batches are below the persistent kernel's whole-tile shapes)."""
    g = golden(name)
    model = build_drmodel(g, arch, gated, dtype="float16")
    for kind in ("p", "q"):
        hidden, reps = model.encode_passage(items_from_golden(g, kind, DEV))
        assert hidden.dtype == torch.float16
        a, b = reps.cpu().double(), torch.from_numpy(g[kind + "_reps"]).double()
        cos = torch.nn.functional.cosine_similarity(a, b, dim=1)
        assert cos.min() > 0.99999, cos
        assert (a - b).abs().max() < 2e-2 * b.abs().max(), (a - b).abs().max()


@pytest.mark.parametrize("L,n", [(24, 40), (48, 24), (128, 8), (200, 6), (256, 4)])
def test_f16_fused_path_tracks_f32_path_across_lengths(L, n):
    """float16 mode on whole-tile shapes (hidden 256: the persistent GEMM with the fused-LayerNorm epilogues and every
    key-tile count of the attention kernel) against the exact-f32 HIP path (itself pinned to the reference fixtures):
    ragged batches, cosine and relative error of the pooled embeddings; the same inputs in bfloat16 for scale."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    torch.manual_seed(7 + L)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=256)
    lm = BertModel(cfg).eval()
    with torch.no_grad():                      # trained-checkpoint-like LayerNorm affines and biases, not the 1 / 0 of an init
        for name, p in lm.named_parameters():
            if "LayerNorm.weight" in name:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
            elif "bias" in name:
                p.copy_(0.1 * torch.randn_like(p))
    rng = np.random.default_rng(L)
    ids, mask = synth_tokens(rng, n, L, vocab=600, lo_len=max(2, L // 3), lo_id=300)
    ids[0, :], mask[0, :] = rng.integers(300, 600, L), 1
    items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    outs = {}
    from openmatch_amd import native as N_
    for dtype in ("float32", "float16", "bfloat16"):
        model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
        N_.check(N_.lib().om_debug_option(19, 0))            # the FUSED path at these row counts (<= 1024 rows default to the few-rows path)
        try:
            hidden, reps = model.encode_passage(items)
        finally:
            N_.check(N_.lib().om_debug_option(19, 1024))
        assert hidden.dtype == {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}[dtype]
        outs[dtype] = reps.double().cpu()
    ref = outs["float32"]
    cos16 = torch.nn.functional.cosine_similarity(outs["float16"], ref, dim=1).min().item()
    cosb = torch.nn.functional.cosine_similarity(outs["bfloat16"], ref, dim=1).min().item()
    rel16 = ((outs["float16"] - ref).abs().max() / ref.abs().max()).item()
    relb = ((outs["bfloat16"] - ref).abs().max() / ref.abs().max()).item()
    print(f"\n[f16 vs f32 path, L={L}, {n * L} tokens] 1 - cos {1 - cos16:.2e} (bf16 {1 - cosb:.2e}); max rel err {rel16:.2e} (bf16 {relb:.2e})")
    assert 1 - cos16 < 5e-6 and rel16 < 5e-3, (cos16, rel16)
    assert (1 - cos16) < 0.25 * (1 - cosb) + 1e-7, (cos16, cosb)


@pytest.mark.parametrize("name,arch,gated", [c for c in CASES if c[1] == "t5"])
def test_float16_request_on_t5_encoder_runs_float16(golden, name, arch, gated, monkeypatch):
    """The reference's `--fp16` is float16 autocast for every backbone (retriever/dense_retriever.py:76): a T5 encoder stack
    (ReLU, and the gated tanh-GELU form) runs the float16 kernels (round 5), closer to the fp32 fixture than the bfloat16 path;
    OM_T5_F16=0 keeps a checkpoint whose activations leave the float16 range on the bfloat16 kernels."""
    g = golden(name)
    model = build_drmodel(g, arch, gated, dtype="float16")
    ref = torch.from_numpy(g["p_reps"]).double()
    hidden, reps = model.encode_passage(items_from_golden(g, "p", DEV))
    assert hidden.dtype == torch.float16
    cos16 = torch.nn.functional.cosine_similarity(reps.cpu().double(), ref, dim=1).min().item()
    monkeypatch.setenv("OM_T5_F16", "0")
    hidden_b, reps_b = model.encode_passage(items_from_golden(g, "p", DEV))
    assert hidden_b.dtype == torch.bfloat16
    cosb = torch.nn.functional.cosine_similarity(reps_b.cpu().double(), ref, dim=1).min().item()
    print(f"\n[{name}] float16: 1 - cos {1 - cos16:.2e}; bfloat16: {1 - cosb:.2e}")
    assert cos16 > 0.99999 and cosb > 0.999
    assert (1 - cos16) < 0.25 * (1 - cosb) + 1e-7


@pytest.mark.parametrize("L,n,act", [(32, 40, "relu"), (128, 8, "relu"), (200, 6, "relu"), (320, 4, "relu"), (128, 8, "gated-gelu")])
def test_t5_f16_fused_path_tracks_f32_path(L, n, act):
    """float16 T5 on whole-tile shapes (d_model 256: the persistent GEMM with the RMSNorm-folded operand, the row-statistics
    epilogue, ReLU; the gated form on the generic tiles; the attention kernels with the relative-position bias, beyond 256
    tokens too) against the exact-f32 HIP path; the same inputs in bfloat16 for scale."""
    from transformers import T5Config, T5EncoderModel
    from openmatch.modeling import DRModelForInference
    torch.manual_seed(11 + L)
    lm = T5EncoderModel(T5Config(d_model=256, d_ff=1024, num_layers=3, num_heads=4, d_kv=64, vocab_size=600, feed_forward_proj=act)).eval()
    with torch.no_grad():
        for name, p in lm.named_parameters():
            if "layer_norm.weight" in name:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
    rng = np.random.default_rng(L)
    ids, mask = synth_tokens(rng, n, L, vocab=600, lo_len=max(2, L // 3), lo_id=300)
    ids[0, :], mask[0, :] = rng.integers(300, 600, L), 1
    items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    outs = {}
    from openmatch_amd import native as N_
    for dtype in ("float32", "float16", "bfloat16"):
        model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=True, dtype=dtype)).to(DEV).eval()
        N_.check(N_.lib().om_debug_option(19, 0))            # the FUSED path at these row counts
        try:
            hidden, reps = model.encode_passage(items)
        finally:
            N_.check(N_.lib().om_debug_option(19, 1024))
        assert hidden.dtype == {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}[dtype]
        outs[dtype] = reps.double().cpu()
    ref = outs["float32"]
    cos16 = torch.nn.functional.cosine_similarity(outs["float16"], ref, dim=1).min().item()
    cosb = torch.nn.functional.cosine_similarity(outs["bfloat16"], ref, dim=1).min().item()
    rel16 = ((outs["float16"] - ref).abs().max() / ref.abs().max()).item()
    relb = ((outs["bfloat16"] - ref).abs().max() / ref.abs().max()).item()
    print(f"\n[T5 {act} f16 vs f32 path, L={L}, {n * L} tokens] 1 - cos {1 - cos16:.2e} (bf16 {1 - cosb:.2e}); max rel err {rel16:.2e} (bf16 {relb:.2e})")
    assert 1 - cos16 < 5e-6 and rel16 < 5e-3, (cos16, rel16)
    assert (1 - cos16) < 0.25 * (1 - cosb) + 1e-7, (cos16, cosb)


def test_autocast_float16_selects_f16_path(golden):
    g = golden("bert_tiny_first")
    model = build_drmodel(g, "bert", False)
    with torch.autocast("cuda", dtype=torch.float16):
        hidden, _ = model.encode_query(items_from_golden(g, "q", DEV))
    assert hidden.dtype == torch.float16


def test_autocast_selects_bf16_path(golden):
    g = golden("bert_tiny_first")
    model = build_drmodel(g, "bert", False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        hidden, _ = model.encode_query(items_from_golden(g, "q", DEV))
    assert hidden.dtype == torch.bfloat16


def test_bert_base_f32_dot_products_within_1e4(golden):
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    g = golden("bert_base_seed0")
    torch.manual_seed(0)
    lm = BertModel(BertConfig()).eval()
    model = DRModelForInference(lm_q=lm, lm_p=lm, model_args=NS(encoder_only=False, dtype="float32")).to(DEV).eval()
    out = model(query=items_from_golden(g, "q", DEV), passage=items_from_golden(g, "p", DEV))
    assert np.abs(out.p_reps.cpu().numpy() - g["p_reps"]).max() < 1e-4
    assert np.abs(out.q_reps.cpu().numpy() - g["q_reps"]).max() < 1e-4
    rel = np.abs((out.q_reps @ out.p_reps.t()).cpu().numpy() - g["scores"]) / np.abs(g["scores"]).max()
    assert rel.max() < 1e-4        # un-normalised CLS vectors: dot products are O(100)


def test_encoder_matches_oracle_on_ragged_batches():
    """Lengths that are not multiples of 32, L=162 (cross-encoder pairs), batch of 1, all-pad tail."""
    from transformers import BertModel
    from openmatch.modeling import DRModelForInference
    from tests.helpers import tiny_bert_config
    torch.manual_seed(3)
    cfg = tiny_bert_config()
    cfg = type(cfg)(**{**cfg.to_dict(), "max_position_embeddings": 256})
    lm = BertModel(cfg).eval()
    sd = {k: v.clone() for k, v in lm.state_dict().items()}      # CPU copy for the oracle
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", normalize=True,
                                model_args=NS(encoder_only=False, dtype="float32")).to(DEV).eval()
    rng = np.random.default_rng(5)
    for B, L in ((1, 7), (3, 50), (2, 162), (5, 33), (2, 256)):
        ids, mask = synth_tokens(rng, B, L, vocab=600, lo_len=min(4, L), lo_id=300)
        items = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        _, ref = encoder_ref.encode(sd, cfg, "bert", items, "mean", None, True)
        _, got = model.encode_passage({k: v.to(DEV) for k, v in items.items()})
        assert np.abs(got.cpu().numpy() - ref.numpy()).max() < 1e-4, (B, L)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_attention_skips_masked_key_tiles_exactly(dtype):
    """Round 4: the 16-bit attention kernel fetches and scores only the 32-key tiles that hold an unmasked key (kmax per batch
    row, omk_mask_extent).  Masks with holes, an unmasked key in the last tile behind a long gap, a single leading key and a
    row without any unmasked key (stays uniform over all L keys, as HF's finfo.min makes it) against the f32 path of the same
    library, which reads every key."""
    from transformers import BertModel
    from openmatch.modeling import DRModelForInference
    from tests.helpers import tiny_bert_config
    torch.manual_seed(11)
    cfg = tiny_bert_config()
    lm = BertModel(cfg).eval()
    m32 = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="float32")).to(DEV).eval()
    m16 = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    B, L = 12, 128
    ids = torch.randint(300, 600, (B, L), generator=g)
    mask = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate((128, 97, 64, 33, 32, 31, 16, 1)):
        mask[b, :n] = 1
    mask[8, :20] = 1; mask[8, 120] = 1                     # a lone key in the last tile behind three masked tiles' worth of gap
    mask[9, ::3] = 1                                         # holes everywhere
    mask[10, 40:50] = 1                                      # leading masked keys
    # row 11: no unmasked key at all
    items = {"input_ids": ids.to(DEV), "attention_mask": mask.to(DEV)}
    h32, _ = m32.encode_passage(items)
    h16, _ = m16.encode_passage(items)
    h32, h16 = h32.float().cpu(), h16.float().cpu()
    assert torch.isfinite(h16).all()
    tol = 2e-2 if dtype == "float16" else 8e-2
    for b in range(B):
        keep = mask[b].bool() if mask[b].any() else torch.ones(L, dtype=torch.bool)      # padded positions carry no contract
        err = (h16[b][keep] - h32[b][keep]).abs().max().item()
        assert err < tol, (dtype, b, err)


@pytest.mark.gpu
@pytest.mark.parametrize("pooling", ["first", "mean"])
@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_packed_rows_encoder_is_bit_identical_to_padded(dtype, pooling):
    """Round 4: om_encoder_forward_packed keeps only the rows up to each sequence's last unmasked token (the reference pads to
    one length and computes over the padding, dataset/data_collator.py:27-38).  Every row of a contraction and every
    sequence of the attention is computed independently of where it sits, so the representations must be the SAME BITS as
    om_encoder_forward's -- ragged lengths, a full-length row, masks with holes, leading masked tokens, a row without any
    token; through the compact collator batch (host-side lengths) and through an explicit row bound.  A bound that is too
    small must poison the batch (NaN), never truncate it."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    from openmatch_amd import encoder as enc_mod
    from openmatch_amd.encoder import compute_dtype_code, hip_encode, packed_rows_bound
    from openmatch_amd.feed import pack_token_batch, token_rows_bound
    torch.manual_seed(5)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=128)
    lm = BertModel(cfg).eval()
    with torch.no_grad():
        for name, p in lm.named_parameters():
            if "LayerNorm.weight" in name:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
            elif "bias" in name:
                p.copy_(0.1 * torch.randn_like(p))
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling=pooling, model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
    code = compute_dtype_code(model.model_args)
    rng = np.random.default_rng(9)
    B, L = 40, 128
    ids, mask = synth_tokens(rng, B, L, vocab=600, lo_len=3, lo_id=300)
    ids[0, :], mask[0, :] = rng.integers(300, 600, L), 1                     # a full-length row
    prefix = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    # (1) the collator's compact batch: lengths on the host -> the packed entry, chosen by DRModel.encode
    compact = pack_token_batch(dict(prefix))
    rows = token_rows_bound(compact)
    assert rows is not None and rows % 256 == 0 and rows < B * L - 256, rows
    dev_items = {k: v.to(DEV) for k, v in prefix.items()}
    padded = hip_encode(model.lm_p, dev_items, pooling, None, False, code, want_hidden=False)[1]
    assert enc_mod.LAST_CALL == {"rows": B * L, "packed": False}
    via_model = model(passage=compact).p_reps
    assert enc_mod.LAST_CALL == {"rows": rows, "packed": True}
    assert torch.isfinite(padded).all()
    assert torch.equal(via_model, padded)
    explicit = hip_encode(model.lm_p, dev_items, pooling, None, False, code, want_hidden=False, packed_rows=rows)[1]
    assert torch.equal(explicit, padded)
    # (2) masks that are not prefixes: holes, leading masked tokens, a lone late token, an empty row
    mask2 = mask.copy()
    mask2[1, :] = 0; mask2[1, ::3] = 1
    mask2[2, :] = 0; mask2[2, 40:50] = 1
    mask2[3, :] = 0; mask2[3, :20] = 1; mask2[3, 120] = 1
    mask2[4, :] = 0
    m2 = torch.from_numpy(mask2)
    rows2 = packed_rows_bound(m2)
    last = [(np.nonzero(r)[0][-1] + 1) if r.any() else L for r in mask2]
    assert rows2 == (sum(last) + 255) // 256 * 256
    items2 = {"input_ids": dev_items["input_ids"], "attention_mask": m2.to(DEV)}
    padded2 = hip_encode(model.lm_p, items2, pooling, None, False, code, want_hidden=False)[1]
    packed2 = hip_encode(model.lm_p, items2, pooling, None, False, code, want_hidden=False, packed_rows=rows2)[1]
    assert torch.equal(packed2, padded2)
    # (3) a bound below the token count: NaN everywhere
    small = hip_encode(model.lm_p, items2, pooling, None, False, code, want_hidden=False, packed_rows=rows2 - 256)[1]
    assert torch.isnan(small).all()


@pytest.mark.gpu
@pytest.mark.parametrize("pooling", ["first", "mean"])
@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_encode_and_forward_return_the_same_representations(dtype, pooling):
    """16-bit `forward()` pooled from an f32 final LayerNorm while `encode_passage()` (hidden states requested)
    pooled from the 16-bit hidden states -- two answers for one input.  Both pool from the f32 normalisation now (the
    reference's autocast returns fp32 from layer_norm either way), and forward() goes through encode_query / encode_passage,
    so a subclass that overrides them (reference signature, no `want_hidden`) is honoured."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    torch.manual_seed(13)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=64)
    lm = BertModel(cfg).eval()
    rng = np.random.default_rng(2)
    outs = []
    for n, L in ((24, 48), (3, 20)):                     # the fused path (>= 512 tokens) and the per-site path
        ids, mask = synth_tokens(rng, n, L, vocab=600, lo_len=4, lo_id=300)
        items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
        model = DRModelForInference(lm_q=lm, lm_p=lm, pooling=pooling, model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
        hidden, reps = model.encode_passage(items)
        fwd = model(passage=items).p_reps
        assert hidden is not None and reps.dtype == torch.float32
        assert torch.equal(reps, fwd), (n, L, (reps - fwd).abs().max().item())
        outs.append(reps)

    class Doubling(DRModelForInference):
        def encode_passage(self, psg):                   # the reference's signature
            hidden, reps = super().encode_passage(psg)
            return hidden, 2.0 * reps
    sub = Doubling(lm_q=lm, lm_p=lm, pooling=pooling, model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
    assert torch.equal(sub(passage=items).p_reps, 2.0 * outs[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["bert", "t5"])
@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_packed_rows_beyond_256_tokens_are_bit_identical_to_padded(dtype, arch):
    """Round 6: the packed-rows encoder takes sequences of up to 1 024 tokens (was 256): the online-softmax attention kernel reads each
    sequence's own rows and length, the mask / position-bias table / padded pitch stay.  Same bits as the padded entry at 384 and 520
    tokens: ragged lengths, a full-length row, a mask with holes, an empty row; BERT and T5 (relative-position bias); both poolings."""
    from transformers import BertConfig, BertModel, T5Config, T5EncoderModel
    from openmatch.modeling import DRModelForInference
    from openmatch_amd import encoder as enc_mod
    from openmatch_amd.encoder import compute_dtype_code, hip_encode, packed_rows_bound
    torch.manual_seed(15)
    if arch == "bert":
        lm = BertModel(BertConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                                  max_position_embeddings=640)).eval()
    else:
        lm = T5EncoderModel(T5Config(d_model=256, d_ff=1024, num_layers=2, num_heads=4, d_kv=64, vocab_size=600, feed_forward_proj="relu")).eval()
    rng = np.random.default_rng(19)
    for L, pooling in ((384, "mean"), (520, "first")):
        model = DRModelForInference(lm_q=lm, lm_p=lm, pooling=pooling, model_args=NS(encoder_only=arch == "t5", dtype=dtype)).to(DEV).eval()
        code = compute_dtype_code(model.model_args)
        B = 12
        ids, mask = synth_tokens(rng, B, L, vocab=600, lo_len=5, lo_id=300)
        ids[0, :], mask[0, :] = rng.integers(300, 600, L), 1
        mask[1, :] = 0; mask[1, ::3] = 1
        mask[2, :] = 0
        m = torch.from_numpy(mask)
        rows = packed_rows_bound(m)
        assert rows is not None and rows < B * L - 256
        items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": m.to(DEV)}
        padded = hip_encode(model.lm_p, items, pooling, None, False, code, want_hidden=False)[1]
        assert enc_mod.LAST_CALL == {"rows": B * L, "packed": False}
        packed = hip_encode(model.lm_p, items, pooling, None, False, code, want_hidden=False, packed_rows=rows)[1]
        assert enc_mod.LAST_CALL == {"rows": rows, "packed": True}, enc_mod.LAST_CALL
        keep = torch.ones(B, dtype=torch.bool); keep[2] = False          # (the empty row: a softmax over no keys)
        assert torch.isfinite(padded[keep]).all()
        assert torch.equal(packed[keep], padded[keep]), (L, (packed[keep] - padded[keep]).abs().max().item())
        if rows - 512 >= 512:          # a bound below the token count: NaN everywhere
            small = hip_encode(model.lm_p, items, pooling, None, False, code, want_hidden=False, packed_rows=rows - 512)[1]
            assert torch.isnan(small).all()


@pytest.mark.gpu
@pytest.mark.parametrize("pooling", ["first", "mean"])
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_packed_rows_t5_encoder_is_bit_identical_to_padded(pooling, dtype):
    """The packed-rows entry on the fused T5 path (GTR shape in miniature: both 16-bit formats, RMSNorm folded into the GEMMs, the
    relative-position bias table read at the sequence's own positions): same bits as the padded entry."""
    from transformers import T5Config, T5EncoderModel
    from openmatch.modeling import DRModelForInference
    from openmatch_amd import encoder as enc_mod
    from openmatch_amd.encoder import compute_dtype_code, hip_encode, packed_rows_bound
    torch.manual_seed(8)
    cfg = T5Config(d_model=256, d_ff=1024, num_layers=3, num_heads=4, d_kv=64, vocab_size=600, feed_forward_proj="relu")
    lm = T5EncoderModel(cfg).eval()
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling=pooling, model_args=NS(encoder_only=True, dtype=dtype)).to(DEV).eval()
    code = compute_dtype_code(model.model_args)
    rng = np.random.default_rng(4)
    B, L = 40, 128
    ids, mask = synth_tokens(rng, B, L, vocab=600, lo_len=3, lo_id=300)
    ids[0, :], mask[0, :] = rng.integers(300, 600, L), 1
    mask[1, :] = 0; mask[1, ::3] = 1                       # holes
    mask[2, :] = 0                                        # an empty row
    m = torch.from_numpy(mask)
    rows = packed_rows_bound(m)
    items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": m.to(DEV)}
    padded = hip_encode(model.lm_p, items, pooling, None, False, code, want_hidden=False)[1]
    assert enc_mod.LAST_CALL == {"rows": B * L, "packed": False}
    packed = hip_encode(model.lm_p, items, pooling, None, False, code, want_hidden=False, packed_rows=rows)[1]
    assert enc_mod.LAST_CALL == {"rows": rows, "packed": True}
    assert torch.isfinite(padded).all() and torch.equal(packed, padded)


@pytest.mark.gpu
def test_cross_encoder_takes_packed_rows_for_compact_pair_batches():
    """RRModel.encode on the collator's compact batch (16-bit ids, one host-side length per pair, token types 0 / 1): the
    packed-rows entry with the bf16 two-plane residual stream and a 1-wide head -- the same scores, bit for bit, as the padded
    tensors give."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import LinearHead, RRModel
    from openmatch_amd import encoder as enc_mod
    from openmatch_amd.feed import pack_token_batch
    torch.manual_seed(23)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=192)
    model = RRModel(lm=BertModel(cfg).eval(), head=LinearHead(256, 1), pooling="first",
                    model_args=NS(encoder_only=False, dtype="bfloat16")).to(DEV).eval()
    rng = np.random.default_rng(6)
    n, L = 24, 162
    ids, mask = synth_tokens(rng, n, L, vocab=600, lo_len=20, lo_id=300)
    tt = np.zeros_like(ids)
    for i in range(n):
        ln = int(mask[i].sum()); tt[i, ln // 3:ln] = 1
    host = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.from_numpy(tt)}
    with torch.no_grad():
        padded = model.encode({k: v.to(DEV) for k, v in host.items()})
        assert enc_mod.LAST_CALL["packed"] is False
        packed = model.encode(pack_token_batch(dict(host)))
        assert enc_mod.LAST_CALL["packed"] is True and enc_mod.LAST_CALL["rows"] < n * L
    assert padded.shape == (n, 1) and torch.isfinite(padded).all() and torch.equal(packed, padded)


def _encode_with_fused_ln(model, items, on):
    """A/B switch of the encoder (include/openmatch_hip.h: om_debug_option(OM_OPT_ENCODER_FUSED_LN, .))."""
    from openmatch_amd import native as N
    N.check(N.lib().om_debug_option(0, int(on)))
    N.check(N.lib().om_debug_option(19, 0))      # both sides on the tile kernels: no few-rows path (OM_OPT_GEMM_SKINNY_M; default 1024 rows)
    try:
        return model.encode_passage(items)
    finally:
        N.check(N.lib().om_debug_option(0, 1))
        N.check(N.lib().om_debug_option(19, 1024))


@pytest.mark.parametrize("hidden,heads,ffn", [(256, 4, 512), (512, 8, 1536)])
def test_fused_layernorm_path_matches_unfused_and_oracle(hidden, heads, ffn):
    """bf16 batches of >= 512 tokens take the path where LayerNorm is folded into the GEMMs around it
    (encoder.hip): same embeddings as the launch-per-LayerNorm path and as the f32 oracle, incl. a
    ragged batch, a non-multiple-of-256 row count and non-trivial LayerNorm affines."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    torch.manual_seed(11)
    # (the fused epilogues take whole 256-column tiles: one and two of them here)
    cfg = BertConfig(hidden_size=hidden, num_hidden_layers=3, num_attention_heads=heads, intermediate_size=ffn,
                     vocab_size=600, max_position_embeddings=128)
    lm = BertModel(cfg).eval()
    with torch.no_grad():                      # random-init LayerNorms are (1, 0): make the affines matter
        for name, p in lm.named_parameters():
            if "LayerNorm.weight" in name:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
            elif "LayerNorm.bias" in name:
                p.copy_(0.2 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in lm.state_dict().items()}
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", normalize=True,
                                model_args=NS(encoder_only=False, dtype="bfloat16")).to(DEV).eval()
    rng = np.random.default_rng(8)
    for B, L in ((8, 64), (9, 77), (40, 128)):
        ids, mask = synth_tokens(rng, B, L, vocab=600, lo_len=5, lo_id=300)
        items = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        dev_items = {k: v.to(DEV) for k, v in items.items()}
        _, ref = encoder_ref.encode(sd, cfg, "bert", items, "mean", None, True)
        _, fused = _encode_with_fused_ln(model, dev_items, 1)
        _, plain = _encode_with_fused_ln(model, dev_items, 0)
        fused, plain, ref = fused.float().cpu(), plain.float().cpu(), ref.float()
        cos = lambda a, b: torch.nn.functional.cosine_similarity(a, b, dim=1).min().item()
        assert cos(plain, ref) > 0.999, (B, L, cos(plain, ref))
        assert cos(fused, ref) > 0.999, (B, L, cos(fused, ref))
        assert cos(fused, plain) > 0.9995, (B, L, cos(fused, plain))
        assert not torch.equal(fused, plain)          # the two paths really are different code


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_residual_stream_variants_of_the_fused_path(dtype):
    """The pre-LayerNorm residual stream of the fused 16-bit path (OM_OPT_ENCODER_TWO_PLANE, include/openmatch_hip.h): one 16-bit plane,
    two 16-bit planes (default; the continuous-ring kernel 7r16 with LNF == 3 for float16 and -- OM_GEMM_CONT bit 9 -- for bfloat16, whose
    default is the restart-per-tile kernel), and float16's 16 + 8-bit form (LNF == 4: e5m2 remainders in the kernel's own lane order,
    decoded by the final LayerNorm through omk_lo8_offset).  Every variant tracks the f32 oracle; the two-plane forms are closer to it
    than the one-plane form, and the eight-bit plane lands where the sixteen-bit one does (first and mean pooling: the CLS-row gather and
    the all-rows read of the final LayerNorm; a ragged batch and a row count that is not a multiple of 256)."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    from openmatch_amd import native as N
    torch.manual_seed(13)
    cfg = BertConfig(hidden_size=512, num_hidden_layers=4, num_attention_heads=8, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=128)
    lm = BertModel(cfg).eval()
    with torch.no_grad():
        for name, p in lm.named_parameters():
            if "LayerNorm.weight" in name:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
            elif "LayerNorm.bias" in name:
                p.copy_(0.2 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in lm.state_dict().items()}
    rng = np.random.default_rng(21)
    bit = 2 if dtype == "float16" else 1
    variants = [("one", 0, 495), ("two16", bit, 495)]
    variants += [("two8", 2 | 4, 495)] if dtype == "float16" else [("two16_cont", 1, 495 | 512)]
    err = {}
    for pooling in ("first", "mean"):
        model = DRModelForInference(lm_q=lm, lm_p=lm, pooling=pooling, normalize=False,
                                    model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
        for B, L in ((40, 128), (21, 100)):
            ids, mask = synth_tokens(rng, B, L, vocab=600, lo_len=5, lo_id=300)
            items = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
            dev_items = {k: v.to(DEV) for k, v in items.items()}
            _, ref = encoder_ref.encode(sd, cfg, "bert", items, pooling, None, False)
            ref = ref.double()
            for name, planes, cont in variants:
                N.check(N.lib().om_debug_option(11, planes)); N.check(N.lib().om_debug_option(16, cont)); N.check(N.lib().om_debug_option(19, 0))
                try:
                    _, reps = model.encode_passage(dev_items)
                    _, again = model.encode_passage(dev_items)
                finally:
                    N.check(N.lib().om_debug_option(11, 3)); N.check(N.lib().om_debug_option(16, 495)); N.check(N.lib().om_debug_option(19, 1024))
                assert torch.isfinite(reps).all() and torch.equal(reps, again), (name, pooling, B, L)
                e = ((reps.double().cpu() - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
                err.setdefault(name, []).append(e)
    worst = {k: max(v) for k, v in err.items()}
    print(f"\n[residual stream, {dtype}] worst relative row error vs the f32 oracle: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    tol = 2e-3 if dtype == "float16" else 2e-2
    assert all(v < tol for v in worst.values()), worst
    assert worst["two16"] < worst["one"], worst
    if dtype == "float16":
        assert worst["two8"] < worst["one"] and worst["two8"] < 1.25 * worst["two16"], worst
    else:
        assert worst["two16_cont"] < worst["one"], worst


@pytest.mark.parametrize("gated", [False, True])
def test_fused_rmsnorm_path_matches_unfused_and_oracle(gated):
    """The T5 counterpart: RMSNorm folded into the GEMMs (bf16, >= 512 tokens), GTR-style tail."""
    from transformers import T5Config, T5EncoderModel
    from openmatch.modeling import DRModelForInference, LinearHead
    torch.manual_seed(12 + gated)
    cfg = T5Config(d_model=256, d_ff=512, num_layers=3, num_heads=4, d_kv=64, vocab_size=600,
                   feed_forward_proj="gated-gelu" if gated else "relu")
    lm = T5EncoderModel(cfg).eval()
    with torch.no_grad():
        for name, p in lm.named_parameters():
            if "layer_norm" in name:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
            elif "relative_attention_bias" in name:
                p.copy_(0.5 * torch.randn_like(p))
    head = LinearHead(256, 256)
    sd = {k: v.clone() for k, v in lm.state_dict().items()}
    hw = head.linear.weight.detach().clone()
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", normalize=True, head_q=head, head_p=head,
                                model_args=NS(encoder_only=True, dtype="bfloat16")).to(DEV).eval()
    rng = np.random.default_rng(9)
    for B, L in ((8, 64), (9, 77), (24, 128)):
        ids, mask = synth_tokens(rng, B, L, vocab=600, lo_len=5, lo_id=3)
        items = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        dev_items = {k: v.to(DEV) for k, v in items.items()}
        _, ref = encoder_ref.encode(sd, cfg, "t5", items, "mean", hw, True)
        _, fused = _encode_with_fused_ln(model, dev_items, 1)
        _, plain = _encode_with_fused_ln(model, dev_items, 0)
        fused, plain, ref = fused.float().cpu(), plain.float().cpu(), ref.float()
        cos = lambda a, b: torch.nn.functional.cosine_similarity(a, b, dim=1).min().item()
        assert cos(plain, ref) > 0.999, (B, L, cos(plain, ref))
        assert cos(fused, ref) > 0.999, (B, L, cos(fused, ref))
        assert cos(fused, plain) > 0.9995, (B, L, cos(fused, plain))
        assert torch.equal(fused, plain) == gated      # gated layers keep the norm kernels (encoder.hip)


# ------------------------------------------------------------------------------- search
def _adjudicate(I_gpu, I_ref, P, Q, k):
    P64, Q64 = torch.from_numpy(P).double(), torch.from_numpy(Q).double()

    def full(q, disputed):
        sc = P64 @ Q64[q]
        kth = torch.topk(sc, min(k, sc.numel())).values[-1].item()
        return sc[torch.tensor(disputed)].numpy(), kth
    return flatip.topk_sets_equal(I_gpu, I_ref, full, rel_tol=2e-6)


@pytest.mark.parametrize("precision", ["f32", "f16_rescore"])
@pytest.mark.parametrize("n,nq,k,clustered", [(1000, 100, 100, True), (50000, 64, 1000, True),
                                               (50000, 17, 10, False), (300, 5, 1000, False),
                                               (50000, 128, 1000, True), (40077, 97, 100, False), (20000, 33, 1000, False),
                                               (30011, 300, 100, True)])      # (> 128 queries, not a multiple of 256: generation 7 over the padded query panel)
def test_flat_ip_search_matches_oracle(precision, n, nq, k, clustered):
    from openmatch_amd.index import FlatIPIndex
    rng = np.random.default_rng(n + k)
    d = 768
    mean = rng.standard_normal(d).astype(np.float32) if clustered else 0
    P = (mean + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    Q = (mean + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
    if clustered:
        P /= np.linalg.norm(P, axis=1, keepdims=True); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    idx = FlatIPIndex(d, device=DEV, precision=precision)
    idx.add(P[: n // 2]); idx.add(P[n // 2:])
    D, I = idx.search(Q, k)
    if precision == "f16_rescore" and n >= k:
        assert idx.last_search_info["scan"] == "f16+rescore", idx.last_search_info   # no silent f32 retry
    ref = flatip.IndexFlatIP(d); ref.add(P)
    Dr, Ir = ref.search(Q, k)
    kk = min(n, k)
    assert (np.diff(D[:, :kk], axis=1) <= 0).all()
    assert (I[:, kk:] == -1).all() and (D[:, kk:] == np.float32(-3.4028235e38)).all()
    n_exact, n_tie, n_bad, detail = _adjudicate(I, Ir, P, Q, k)
    print(f"[{precision}] N={n} k={k}: id sets identical for {n_exact}/{nq}, near-tie only {n_tie}, wrong {n_bad}")
    assert n_bad == 0, detail
    # returned scores are the exact f32 inner products of the returned ids
    exact = np.einsum("qd,qkd->qk", Q.astype(np.float64), P[np.clip(I[:, :kk], 0, None)].astype(np.float64))
    assert np.abs(D[:, :kk] - exact).max() < 1e-4 * max(1.0, np.abs(exact).max())


@pytest.mark.parametrize("precision", ["f32", "f16_rescore"])
@pytest.mark.parametrize("nq", [4, 64, 128])
def test_flat_ip_search_with_heavily_duplicated_rows(precision, nq):
    """Collisions: 60 000 rows that are 40 distinct vectors, 1 500 shuffled copies each.  Every score level is a 1 500-way
    exact tie, so the filtered scan's candidate lists fill with ties (longer than the selection kernel's LDS copy, then
    past the list capacity) and the fast schedule has to hand over to the step-by-step path.  The result must still be
    exact: scores equal to the oracle's, ids any members of the tied groups (fp64-adjudicated)."""
    from openmatch_amd.index import FlatIPIndex
    rng = np.random.default_rng(11 + nq)
    d, groups, copies, k = 768, 40, 1500, 1000
    base = rng.standard_normal((groups, d)).astype(np.float32)
    P = np.repeat(base, copies, axis=0)[rng.permutation(groups * copies)]
    Q = (base[rng.integers(0, groups, nq)] + 0.5 * rng.standard_normal((nq, d))).astype(np.float32)
    idx = FlatIPIndex(d, device=DEV, precision=precision)
    idx.add(P)
    D, I = idx.search(Q, k)
    ref = flatip.IndexFlatIP(d); ref.add(P)
    Dr, Ir = ref.search(Q, k)
    assert (np.diff(D, axis=1) <= 0).all()
    assert np.abs(D - Dr).max() <= 1e-4 * np.abs(Dr).max(), np.abs(D - Dr).max()
    n_exact, n_tie, n_bad, detail = _adjudicate(I, Ir, P, Q, k)
    print(f"[{precision}, duplicated rows, Q={nq}] id sets identical for {n_exact}/{nq}, tie-only {n_tie}, wrong {n_bad}; {idx.last_search_info}")
    assert n_bad == 0, detail
    assert all(len(set(row.tolist())) == k for row in I)          # no id returned twice


def test_search_properties_at_scale():
    """Size-independent properties on a 1M x 768 shard (too big for the CPU oracle in seconds):
    every row retrieves itself first; results are sorted; both precisions return the same ids."""
    from openmatch_amd.index import FlatIPIndex
    g = torch.Generator(device=DEV).manual_seed(11)
    n, d, k = 1_000_000, 768, 1000
    P = torch.randn(n, d, device=DEV, generator=g)
    P = torch.nn.functional.normalize(P + 0.5 * torch.randn(1, d, device=DEV, generator=g), dim=1)
    probe = torch.arange(0, n, n // 256, device=DEV)[:256]
    out = {}
    for precision in ("f32", "f16_rescore"):
        idx = FlatIPIndex(d, device=DEV, precision=precision)
        idx.add(P)
        D, I = idx.search_device(P[probe], k)
        assert (I[:, 0] == probe).all()                       # <p,p> = 1 is the unique maximum
        assert (D[:, 1:] <= D[:, :-1]).all()
        assert (I >= 0).all() and (I < n).all()
        assert all(len(set(row.tolist())) == k for row in I[:8].cpu())
        if precision == "f16_rescore":
            assert idx.last_search_info["scan"] == "f16+rescore", idx.last_search_info
        out[precision] = (D.cpu(), I.cpu())
    # both scans against an independent device-side check (chunked torch fp32 matmul + topk); differing id sets are
    # adjudicated in fp64: every disputed id must tie with the k-th score (oracle/device_check.py)
    from oracle import device_check
    Dr, Ir = device_check.reference_topk(P, P[probe], k)
    for precision in ("f32", "f16_rescore"):
        n_exact, n_tie, n_bad, detail = device_check.adjudicate(P, P[probe], out[precision][1], Ir, k)
        print(f"[{precision}, 1M rows] id sets identical for {n_exact}/{len(probe)}, fp64 near-tie only {n_tie}, wrong {n_bad}")
        assert n_bad == 0, detail
        assert (out[precision][0] - Dr.cpu()).abs().max() <= 1e-4


def test_sharded_topk_over_a_one_rank_rccl_group_equals_the_local_search():
    """The candidate exchange of the sharded search (index.sharded_topk: ONE all_to_all_single of score bits | shard-local
    ids | the sender's offset, round 4) on GPU tensors through RCCL -- a one-rank group is all a one-GPU box offers; the
    arithmetic across ranks is covered over gloo at world size 2 and 8 (tests/test_distributed_cpu.py).  An id offset
    beyond 2^31 checks the 64-bit offset's trip through the int32 payload."""
    import torch.distributed as dist
    from openmatch_amd.index import FlatIPIndex, sharded_topk
    g = torch.Generator(device=DEV).manual_seed(3)
    n, d, nq, k = 30000, 768, 77, 100
    P = torch.randn(n, d, device=DEV, generator=g)
    Q = torch.randn(nq, d, device=DEV, generator=g)
    idx = FlatIPIndex(d, device=DEV, precision="f16_rescore")
    idx.add(P)
    offset = (1 << 33) + 12345
    D0, I0 = idx.search_device(Q, k, id_offset=offset)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29547", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        D1, I1, blk = sharded_topk(idx, Q, k, offset)
    finally:
        dist.destroy_process_group()
    assert blk == nq
    assert torch.equal(D1[:nq], D0) and torch.equal(I1[:nq], I0)
    assert int(I1.min()) >= offset


def test_topk_merge_equals_single_index():
    """Per-shard search + om_topk_merge == one index over all rows (K14 / faiss shard merge)."""
    from openmatch_amd.index import FlatIPIndex, merge_topk
    rng = np.random.default_rng(2)
    n, d, nq, k, W = 40000, 128, 33, 200, 8
    P = rng.standard_normal((n, d)).astype(np.float32); Q = rng.standard_normal((nq, d)).astype(np.float32)
    whole = FlatIPIndex(d, device=DEV, precision="f32"); whole.add(P)
    Dw, Iw = whole.search_device(torch.from_numpy(Q), k)
    parts = []
    for w in range(W):
        sh = FlatIPIndex(d, device=DEV, precision="f32"); sh.add(P[w * n // W:(w + 1) * n // W])
        parts.append(sh.search_device(torch.from_numpy(Q), k, id_offset=w * n // W))
    Dm, Im = merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)
    assert (Im == Iw).all() and (Dm == Dw).all()


def test_retriever_end_to_end_on_reference_fixture(golden, tmp_path):
    """Config 1 plumbing: shard pickles -> Retriever.from_embeddings -> search -> TREC -> MRR@10,
    against what the reference produced on the same embeddings."""
    import pickle
    from openmatch.retriever import Retriever
    from openmatch.utils import eval_mrr, load_from_trec, save_as_trec
    g = golden("retrieval_1k")
    doc_ids, qry_ids = list(g["doc_ids"]), list(g["qry_ids"])
    for r in range(2):
        with open(tmp_path / f"embeddings.corpus.rank.{r}", "wb") as f:
            pickle.dump((g["P"][r * 500:(r + 1) * 500], doc_ids[r * 500:(r + 1) * 500]), f, protocol=4)
        with open(tmp_path / f"embeddings.query.rank.{r}", "wb") as f:
            pickle.dump((g["Q"][r * 50:(r + 1) * 50], qry_ids[r * 50:(r + 1) * 50]), f, protocol=4)
    args = NS(device=DEV, output_dir=str(tmp_path), world_size=1, process_index=0, local_process_index=0, fp16=False)
    retriever = Retriever.from_embeddings(torch.nn.Linear(1, 1), args)
    args.world_size = 2            # two query shard files, as two encoding ranks leave them
    retriever._sharded = False
    run = retriever.search(100)
    pos = {d: i for i, d in enumerate(doc_ids)}
    I_run = np.array([[pos[d] for d in run[q]] for q in qry_ids], np.int64)
    n_exact, n_tie, n_bad, detail = _adjudicate(I_run, np.asarray(g["I"], np.int64), np.asarray(g["P"], np.float32), np.asarray(g["Q"], np.float32), 100)
    print(f"retriever fixture: id sets identical for {n_exact}/100, fp64 near-tie only {n_tie}, wrong {n_bad}")
    assert n_bad == 0, detail
    save_as_trec(run, str(tmp_path / "run.trec"))
    back = load_from_trec(str(tmp_path / "run.trec"))
    qrel = {q: {d: 1} for q, d in zip(qry_ids, g["qrel_docs"])}
    assert abs(eval_mrr(qrel, back, cutoff=10)["all"] - float(g["mrr10"])) < 1e-4


# ------------------------------------------------------------------------------- loss
def test_contrastive_loss_matches_reference_fixture(golden):
    from openmatch_amd.ops import contrastive_loss
    g = golden("train_bert_tiny")
    q = torch.from_numpy(g["q_reps"]).to(DEV).requires_grad_()
    p = torch.from_numpy(g["p_reps"]).to(DEV).requires_grad_()
    loss, scores = contrastive_loss(q, p, int(g["n_psg"]), 1.0, q, 0, p, 0)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert np.abs(scores.cpu().numpy() - g["scores"]).max() < 1e-5
    loss.backward()
    qc = torch.from_numpy(g["q_reps"]).requires_grad_(); pc = torch.from_numpy(g["p_reps"]).requires_grad_()
    lref, _ = retrieval_ref.contrastive_loss(qc, pc, int(g["n_psg"]))
    lref.backward()
    assert (q.grad.cpu() - qc.grad).abs().max() < 1e-6 and (p.grad.cpu() - pc.grad).abs().max() < 1e-6


def test_loss_objects_full_call_surface(golden):
    """SimpleContrastiveLoss / DistributedContrastiveLoss as OBJECTS with the reference's call surface (loss.py:7-38):
    default in-batch target, an explicit `target=` (incl. torch's ignore_index), `reduction=` mean / sum / none --
    values and gradients against F.cross_entropy on the CPU.  The distributed variant runs on a 1-rank group."""
    import torch.distributed as dist
    import torch.nn.functional as F
    from openmatch.loss import DistributedContrastiveLoss, SimpleContrastiveLoss
    g = golden("train_bert_tiny")
    q0, p0 = torch.from_numpy(g["q_reps"]), torch.from_numpy(g["p_reps"])          # [4,128], [8,128]
    explicit = torch.tensor([1, 0, 7, -100])
    weights = torch.tensor([0.5, -1.0, 2.0, 0.25])

    def reference(target, reduction, scale=1.0):
        q, p = q0.clone().requires_grad_(), p0.clone().requires_grad_()
        tgt = torch.arange(0, 4 * 2, 2) if target is None else target
        loss = F.cross_entropy(q @ p.t(), tgt, reduction=reduction) * scale
        (loss * weights).sum().backward() if reduction == "none" else loss.backward()
        return loss.detach(), q.grad, p.grad

    def check(fn, scale=1.0):
        for target in (None, explicit):
            for reduction in ("mean", "sum", "none"):
                q, p = q0.clone().to(DEV).requires_grad_(), p0.clone().to(DEV).requires_grad_()
                kw = {}
                if target is not None:
                    kw["target"] = target.to(DEV)
                if reduction != "mean":
                    kw["reduction"] = reduction
                loss = fn(q, p, **kw)
                (loss * weights.to(DEV)).sum().backward() if reduction == "none" else loss.backward()
                l_ref, gq, gp = reference(target, reduction, scale)
                assert loss.shape == l_ref.shape
                assert (loss.detach().cpu() - l_ref).abs().max() < 1e-5, (target, reduction)
                assert (q.grad.cpu() - gq).abs().max() < 1e-6 and (p.grad.cpu() - gp).abs().max() < 1e-6, (target, reduction)
    check(SimpleContrastiveLoss())
    with pytest.raises(ValueError):
        SimpleContrastiveLoss()(q0.to(DEV), p0.to(DEV), reduction="median")
    for bad in (p0.shape[0], -1):                        # F.cross_entropy raises on a class index outside [0, C)
        t_bad = torch.zeros(q0.shape[0], dtype=torch.int64); t_bad[1] = bad
        with pytest.raises(IndexError):
            SimpleContrastiveLoss()(q0.to(DEV), p0.to(DEV), target=t_bad)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        check(DistributedContrastiveLoss())                    # world size 1: gather is the identity, scale 1
        check(DistributedContrastiveLoss(scale_loss=False))
    finally:
        dist.destroy_process_group()


def test_native_rccl_entry_points_single_rank():
    """The RCCL wrappers of the C ABI (om_comm_*, om_allgather_rows, om_allreduce_grads, om_exchange_topk) on a
    1-rank communicator: every collective degenerates to the identity, which checks the run-time binding of RCCL,
    the unique-id bootstrap and the argument plumbing on the one GPU this box has (2 ranks: tests/test_multigpu.py)."""
    from openmatch_amd.comm import RcclComm
    comm = RcclComm(1, 0, DEV, RcclComm.unique_id())
    try:
        x = torch.randn(5, 7, device=DEV)
        assert torch.equal(comm.allgather_rows(x), x)
        g = torch.randn(1000, device=DEV)
        g0 = g.clone()
        comm.allreduce_grads_(g, average=True)
        assert torch.allclose(g, g0)
        D, I = torch.randn(6, 4, device=DEV), torch.randint(0, 99, (6, 4), device=DEV)
        rD, rI = comm.exchange_topk(D, I)
        torch.cuda.synchronize()
        assert rD.shape == (1, 6, 4) and torch.equal(rD[0], D) and torch.equal(rI[0], I)
    finally:
        comm.close()


# ------------------------------------------------------------------------------- training
def _train_model(g, dtype="float32", p_drop=0.0):
    from openmatch.modeling import DRModel, LinearHead
    cfg, lm = model_from_golden(g, "bert", hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop)
    head = LinearHead(128, 128)
    head.linear.weight.data.copy_(torch.from_numpy(g["head_w"]))
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                    model_args=NS(encoder_only=False, dtype=dtype), data_args=NS(train_n_passages=int(g["n_psg"])),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=4))
    return model.to(DEV).train()


def _train_batch(g):
    mk = lambda k: {"input_ids": torch.from_numpy(g[k + "_input_ids"]).to(DEV),
                    "attention_mask": torch.from_numpy(g[k + "_attention_mask"]).to(DEV),
                    "token_type_ids": torch.zeros_like(torch.from_numpy(g[k + "_input_ids"])).to(DEV)}
    return mk("q"), mk("p")


def test_training_step_f32_matches_reference_gradients(golden):
    """DRModel.forward(train) + loss.backward() through the HIP encoder backward vs the reference's
    autograd (dropout 0): loss within 1e-5, every parameter gradient within 1e-3 relative L2 and
    2e-5 absolute of the reference's."""
    g = golden("train_bert_tiny")
    model = _train_model(g)
    q, p = _train_batch(g)
    out = model(query=q, passage=p)
    assert abs(out.loss.item() - float(g["loss"])) < 1e-5
    assert np.abs(out.scores.detach().cpu().numpy() - g["scores"]).max() < 1e-5
    out.loss.backward()
    worst = ("", 0.0)
    names = dict(model.lm_q.named_parameters())
    checked = 0
    for key in g.files:
        if not key.startswith("g::"):
            continue
        name = key[3:]
        ref = torch.from_numpy(g[key])
        got = (model.head_q.linear.weight.grad if name == "head_w" else names[name].grad)
        assert got is not None, name
        got = got.cpu()
        rel = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
        amax = (got - ref).abs().max().item()
        # (the key-bias gradient is identically zero -- softmax is shift invariant -- so only its
        # absolute size can be checked)
        assert (rel < 1e-3 or amax < 1e-7) and amax < 2e-5, (name, rel, amax)
        if amax >= 1e-7:
            worst = max(worst, (name, rel), key=lambda t: t[1])
        checked += 1
    assert checked > 30
    print("worst relative gradient error:", worst)


@pytest.mark.parametrize("gated", [False, True])
def test_t5_training_step_matches_reference_gradients(golden, gated):
    """GTR-style T5 encoder (mean pooling, head, normalise): loss and EVERY parameter gradient --
    shared embedding, RMSNorm weights, relative-position bias table, q/k/v/o, wi (/wi_0, wi_1), wo --
    against the reference's autograd (tests/golden/train_t5_tiny_*.npz, dropout 0, L = 100 passages)."""
    from openmatch.modeling import DRModel, LinearHead
    g = golden("train_t5_tiny_gated" if gated else "train_t5_tiny_relu")
    cfg, lm = model_from_golden(g, "t5", gated=gated)
    lm.config.dropout_rate = 0.0
    head = LinearHead(128, 128)
    head.linear.weight.data.copy_(torch.from_numpy(g["head_w"]))
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                    model_args=NS(encoder_only=True, dtype="float32"), data_args=NS(train_n_passages=int(g["n_psg"])),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=4)).to(DEV).train()
    mk = lambda k: {"input_ids": torch.from_numpy(g[k + "_input_ids"]).to(DEV),
                    "attention_mask": torch.from_numpy(g[k + "_attention_mask"]).to(DEV)}
    out = model(query=mk("q"), passage=mk("p"))
    assert abs(out.loss.item() - float(g["loss"])) < 1e-5
    assert np.abs(out.scores.detach().cpu().numpy() - g["scores"]).max() < 1e-5
    out.loss.backward()
    names = dict(model.lm_q.named_parameters())
    checked, worst = 0, ("", 0.0)
    for key in g.files:
        if not key.startswith("g::"):
            continue
        name = key[3:]
        ref = torch.from_numpy(g[key])
        got = model.head_q.linear.weight.grad if name == "head_w" else names[name].grad
        assert got is not None, name
        got = got.cpu()
        rel = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
        amax = (got - ref).abs().max().item()
        assert (rel < 1e-3 or amax < 1e-7) and amax < 2e-5, (name, rel, amax)
        worst = max(worst, (name, rel), key=lambda t: t[1])
        checked += 1
    assert checked == (22 if gated else 20), checked
    print("worst relative gradient error:", worst)


def test_training_step_bf16_and_dropout_are_sane(golden):
    g = golden("train_bert_tiny")
    q, p = _train_batch(g)
    # bf16 compute, no dropout: gradients close to the f32 reference (cosine per tensor)
    model = _train_model(g, dtype="bfloat16")
    out = model(query=q, passage=p)
    assert abs(out.loss.item() - float(g["loss"])) < 2e-2
    out.loss.backward()
    names = dict(model.lm_q.named_parameters())
    for key in ("g::encoder.layer.0.attention.self.query.weight", "g::encoder.layer.1.output.dense.weight",
                "g::embeddings.word_embeddings.weight", "g::encoder.layer.0.intermediate.dense.bias"):
        ref = torch.from_numpy(g[key]).flatten().double()
        got = names[key[3:]].grad.cpu().flatten().double()
        cos = torch.dot(ref, got) / (ref.norm() * got.norm())
        assert cos > 0.99, (key, cos.item())
    # dropout 0.1: finite loss / gradients, different masks on different calls, eval() is deterministic
    model = _train_model(g, p_drop=0.1)
    l1 = model(query=q, passage=p).loss
    l2 = model(query=q, passage=p).loss
    assert torch.isfinite(l1) and torch.isfinite(l2) and l1.item() != l2.item()
    l1.backward()
    assert all(torch.isfinite(p_.grad).all() for p_ in model.lm_q.parameters() if p_.grad is not None)
    model.eval()
    with torch.no_grad():
        a = model(query=q, passage=p).loss.item(); b = model(query=q, passage=p).loss.item()
    assert a == b


def test_t5_training_bf16_and_dropout_are_sane(golden):
    """T5 training in bf16 tracks the f32 reference gradients; with dropout_rate 0.1 the step stays finite,
    masks change from call to call, and one optimiser-free descent step along -grad lowers the loss."""
    from openmatch.modeling import DRModel, LinearHead
    g = golden("train_t5_tiny_gated")

    def build(dtype, p_drop):
        cfg, lm = model_from_golden(g, "t5", gated=True)
        lm.config.dropout_rate = p_drop
        head = LinearHead(128, 128)
        head.linear.weight.data.copy_(torch.from_numpy(g["head_w"]))
        return DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                       model_args=NS(encoder_only=True, dtype=dtype), data_args=NS(train_n_passages=int(g["n_psg"])),
                       train_args=NS(negatives_x_device=False, per_device_train_batch_size=4)).to(DEV).train()
    mk = lambda k: {"input_ids": torch.from_numpy(g[k + "_input_ids"]).to(DEV),
                    "attention_mask": torch.from_numpy(g[k + "_attention_mask"]).to(DEV)}
    q, p = mk("q"), mk("p")
    model = build("bfloat16", 0.0)
    out = model(query=q, passage=p)
    assert abs(out.loss.item() - float(g["loss"])) < 3e-2
    out.loss.backward()
    names = dict(model.lm_q.named_parameters())
    for key in ("g::encoder.block.0.layer.0.SelfAttention.q.weight", "g::encoder.block.1.layer.1.DenseReluDense.wo.weight",
                "g::encoder.block.0.layer.1.DenseReluDense.wi_1.weight", "g::encoder.final_layer_norm.weight"):
        ref = torch.from_numpy(g[key]).flatten().double()
        got = names[key[3:]].grad.cpu().flatten().double()
        cos = torch.dot(ref, got) / (ref.norm() * got.norm())
        assert cos > 0.99, (key, cos.item())
    model = build("float32", 0.1)
    l1 = model(query=q, passage=p).loss
    l2 = model(query=q, passage=p).loss
    assert torch.isfinite(l1) and torch.isfinite(l2) and l1.item() != l2.item()
    l1.backward()
    params = [p_ for p_ in list(model.lm_q.parameters()) + list(model.head_q.parameters()) if p_.grad is not None]
    assert all(torch.isfinite(p_.grad).all() for p_ in params)
    model.eval()
    with torch.no_grad():
        before = model(query=q, passage=p).loss.item()
        for p_ in params:
            p_.add_(p_.grad, alpha=-0.05)
        after = model(query=q, passage=p).loss.item()
    assert after < before, (before, after)


@pytest.mark.parametrize("fixture,gated", [("train_t5_tiny_relu", False), ("train_t5_tiny_gated", True)])
def test_t5_training_float16_inside_reference_float16_autocast(golden, fixture, gated, monkeypatch):
    """float16 T5 training (round 6; the reference's `--fp16` is float16 autocast + GradScaler for every backbone,
    trainer/dense_trainer.py:141-149): the float16 kernels' step on the reference-executed fixtures, loss scaled by 1024 as under a
    GradScaler.  Yardstick: the reference's OWN float16-autocast gradients on the same fixture (oracle/make_golden_t5_train.py
    --add-f16: per-tensor relative L2 from its fp32 gradients) -- every tensor within 3 x its yardstick (or 3 x the median
    yardstick where a tensor's own is tiny), the loss within 3 x the reference's own loss deviation; bfloat16 on the same fixture
    is printed beside it."""
    monkeypatch.delenv("OM_TRAIN_F16", raising=False); monkeypatch.delenv("OM_T5_F16", raising=False)
    from openmatch.modeling import DRModel, LinearHead
    from openmatch_amd import native as N
    from openmatch_amd import train as T
    g = golden(fixture)
    yard = {k[len("ac16rel::"):]: float(g[k]) for k in g.files if k.startswith("ac16rel::")}
    assert yard, "fixture without float16-autocast yardsticks (oracle/make_golden_t5_train.py --add-f16)"
    med = float(np.median(list(yard.values())))

    def step(dtype):
        cfg, lm = model_from_golden(g, "t5", gated=gated)
        lm.config.dropout_rate = 0.0
        head = LinearHead(128, 128)
        head.linear.weight.data.copy_(torch.from_numpy(g["head_w"]))
        model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                        model_args=NS(encoder_only=True, dtype=dtype), data_args=NS(train_n_passages=int(g["n_psg"])),
                        train_args=NS(negatives_x_device=False, per_device_train_batch_size=4)).to(DEV).train()
        mk = lambda k: {"input_ids": torch.from_numpy(g[k + "_input_ids"]).to(DEV),
                        "attention_mask": torch.from_numpy(g[k + "_attention_mask"]).to(DEV)}
        out = model(query=mk("q"), passage=mk("p"))
        (out.loss * 1024.0).backward()
        names = dict(lm.named_parameters())
        rel = {}
        for key in yard:
            got = (head.linear.weight.grad if key == "head_w" else names[key].grad).float().cpu().double() / 1024.0
            ref = torch.from_numpy(g["g::" + key]).double()
            rel[key] = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        return out.loss.item(), rel
    loss16, rel16 = step("float16")
    assert T.LAST_TRAIN_CODE == N.OM_F16, "the float16 request must run the float16 kernels"
    lossb, relb = step("bfloat16")
    worst = max(rel16, key=lambda k: rel16[k] / max(yard[k], med))
    print(f"\n[T5 {'gated-gelu' if gated else 'relu'} float16 training] rel-L2 vs the reference's fp32 gradients: median {np.median(list(rel16.values())):.2e} "
          f"(reference float16 autocast {med:.2e}; bfloat16 kernels {np.median(list(relb.values())):.2e}); worst tensor {rel16[worst] / max(yard[worst], med):.2f} x "
          f"its yardstick ({worst}); loss {loss16:.5f} vs fp32 {float(g['loss']):.5f} (reference float16 autocast {float(g['ac16_loss']):.5f})")
    for k in yard:
        assert rel16[k] <= 3.0 * max(yard[k], med), (k, rel16[k], yard[k], med)
    assert abs(loss16 - float(g["loss"])) <= max(3.0 * abs(float(g["ac16_loss"]) - float(g["loss"])), 1e-3)
    assert np.median(list(rel16.values())) < np.median(list(relb.values()))       # three more mantissa bits than bfloat16


def test_t5_decoder_position_float16_training_and_inference_track_f32(monkeypatch):
    """The T5 decoder position (DRModel with encoder_only=False, monoT5) in float16 (round 6): the inference step and the training
    pair on the float16 kernels against the same model in float32 -- representations within 2e-3 relative, every gradient's cosine
    > 0.999 (bfloat16: > 0.99 in the test above), closer than bfloat16 on the median."""
    monkeypatch.delenv("OM_TRAIN_F16", raising=False); monkeypatch.delenv("OM_T5_F16", raising=False)      # (A/B switches that send float16 to the bfloat16 kernels)
    import copy
    from transformers import T5Config, T5ForConditionalGeneration
    from openmatch.modeling import DRModel
    torch.manual_seed(123)
    cfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, d_kv=64, vocab_size=600,
                   feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0, dropout_rate=0.0)
    lm = T5ForConditionalGeneration(cfg)
    rng = np.random.default_rng(23)
    ids, mask = synth_tokens(rng, 8, 64, vocab=600, lo_len=10, lo_id=300)
    items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    R = torch.randn(8, 128, generator=torch.Generator().manual_seed(8)).to(DEV)

    def run(dtype):
        m = copy.deepcopy(lm)
        dr = DRModel(lm_q=m, lm_p=m, model_args=NS(encoder_only=False, dtype=dtype), data_args=NS(train_n_passages=1),
                     train_args=NS(negatives_x_device=False)).to(DEV).train()
        _h, reps = dr.encode_passage(items)
        (reps * R * 256.0).sum().backward()
        grads = {n: p.grad.float().cpu() / 256.0 for n, p in m.named_parameters() if p.grad is not None}
        dr.eval()
        with torch.no_grad():
            _h, inf = dr.encode_passage(items)
        return reps.detach().float().cpu(), inf.float().cpu(), grads
    r32, i32, g32 = run("float32")
    r16, i16, g16 = run("float16")
    rb, ib, gb = run("bfloat16")
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(f"\n[T5 decoder position] training reps rel err float16 {rel(r16, r32):.2e} (bfloat16 {rel(rb, r32):.2e}); inference float16 {rel(i16, i32):.2e} (bfloat16 {rel(ib, i32):.2e})")
    assert rel(r16, r32) < 2e-3 and rel(i16, i32) < 2e-3 and rel(r16, r32) < rel(rb, r32) and rel(i16, i32) < rel(ib, i32)
    cos16, cosb = [], []
    for key, a in g32.items():
        if float(a.abs().max()) == 0.0:
            continue
        c = lambda x: (torch.dot(a.flatten().double(), x.flatten().double()) / (a.norm().double() * x.norm().double())).item()
        cos16.append(c(g16[key])); cosb.append(c(gb[key]))
        assert cos16[-1] > 0.999, (key, cos16[-1])
    assert np.median(cos16) >= np.median(cosb)


def test_training_gradients_match_oracle_autograd_at_bert_width():
    """One bert-base-WIDTH layer (H=768, F=3072, 12 heads), 8 x 128-token passages + 4 x 32-token
    queries: exercises the 256-row GEMM tiles, the fused GELU' epilogue and the wgrad path at real
    shapes.  Reference = torch autograd through the CPU oracle (oracle/encoder_ref.py)."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    torch.manual_seed(11)
    cfg = BertConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
                     vocab_size=600, max_position_embeddings=160, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    ref_lm = BertModel(cfg); ref_lm.load_state_dict(lm.state_dict())
    rng = np.random.default_rng(4)
    p_ids, p_mask = synth_tokens(rng, 8, 128, vocab=600, lo_len=16, lo_id=300)
    q_ids, q_mask = synth_tokens(rng, 4, 32, vocab=600, lo_len=4, lo_id=300)
    tens = lambda a: torch.from_numpy(a)
    # reference on CPU
    sd = dict(ref_lm.named_parameters())
    sd.update({k: v for k, v in ref_lm.named_buffers()})
    hq = encoder_ref.encode(sd, cfg, "bert", {"input_ids": tens(q_ids), "attention_mask": tens(q_mask)}, "first")[1]
    hp = encoder_ref.encode(sd, cfg, "bert", {"input_ids": tens(p_ids), "attention_mask": tens(p_mask)}, "first")[1]
    loss_ref, _ = retrieval_ref.contrastive_loss(hq, hp, 2)
    loss_ref.backward()
    model = DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="float32"),
                    data_args=NS(train_n_passages=2),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=4)).to(DEV).train()
    out = model(query={"input_ids": tens(q_ids).to(DEV), "attention_mask": tens(q_mask).to(DEV)},
                passage={"input_ids": tens(p_ids).to(DEV), "attention_mask": tens(p_mask).to(DEV)})
    assert abs(out.loss.item() - loss_ref.item()) < 1e-3 * max(1.0, abs(loss_ref.item()))
    out.loss.backward()
    got = dict(lm.named_parameters())
    worst = ("", 0.0)
    for name, p in ref_lm.named_parameters():
        if p.grad is None:
            continue
        gg = got[name].grad.cpu()
        rel = ((gg - p.grad).norm() / (p.grad.norm() + 1e-20)).item()
        amax = (gg - p.grad).abs().max().item()
        assert rel < 2e-3 or amax < 1e-7, (name, rel, amax)
        if amax >= 1e-7:
            worst = max(worst, (name, rel), key=lambda t: t[1])
    print("worst relative gradient error at bert width:", worst)


@pytest.mark.parametrize("L,dtype", [(162, "float32"), (192, "float32"), (256, "bfloat16")])
def test_training_gradients_beyond_128_tokens(L, dtype):
    """The training path at the cross-encoder's default pair length (q_max_len + p_max_len + 2 = 162, what
    PairCollator pads to), at the float32 limit (192) and at the 256-token limit (16-bit compute only: the backward
    attention kernel's LDS images): loss and every parameter gradient against torch autograd through the CPU oracle.
    (Round 1 rejected L > 128, so train_rr failed with its default arguments.)"""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    torch.manual_seed(5)
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                     max_position_embeddings=256, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    ref_lm = BertModel(cfg); ref_lm.load_state_dict(lm.state_dict())
    rng = np.random.default_rng(L)
    p_ids, p_mask = synth_tokens(rng, 6, L, vocab=600, lo_len=L // 2, lo_id=300)
    q_ids, q_mask = synth_tokens(rng, 3, 32, vocab=600, lo_len=4, lo_id=300)
    tens = lambda a: torch.from_numpy(a)
    sd = dict(ref_lm.named_parameters())
    sd.update({k: v for k, v in ref_lm.named_buffers()})
    hq = encoder_ref.encode(sd, cfg, "bert", {"input_ids": tens(q_ids), "attention_mask": tens(q_mask)}, "mean")[1]
    hp = encoder_ref.encode(sd, cfg, "bert", {"input_ids": tens(p_ids), "attention_mask": tens(p_mask)}, "mean")[1]
    loss_ref, _ = retrieval_ref.contrastive_loss(hq, hp, 2)
    loss_ref.backward()
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=False, dtype=dtype),
                    data_args=NS(train_n_passages=2),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=3)).to(DEV).train()
    out = model(query={"input_ids": tens(q_ids).to(DEV), "attention_mask": tens(q_mask).to(DEV)},
                passage={"input_ids": tens(p_ids).to(DEV), "attention_mask": tens(p_mask).to(DEV)})
    exact = dtype == "float32"
    assert abs(out.loss.item() - loss_ref.item()) < (1e-4 if exact else 2e-2) * max(1.0, abs(loss_ref.item()))
    out.loss.backward()
    got = dict(lm.named_parameters())
    for name, p in ref_lm.named_parameters():
        if p.grad is None:
            continue
        gg = got[name].grad.cpu().float()
        if exact:
            rel = ((gg - p.grad).norm() / (p.grad.norm() + 1e-20)).item()
            amax = (gg - p.grad).abs().max().item()
            assert rel < 2e-3 or amax < 1e-7, (L, name, rel, amax)
        elif p.grad.norm() > 1e-6:                     # 16-bit compute: direction of every gradient tensor
            cos = (torch.dot(gg.flatten().double(), p.grad.flatten().double()) / (gg.norm().double() * p.grad.norm().double())).item()
            assert cos > 0.98, (L, name, cos)


@pytest.mark.parametrize("M,N,K,with_bias", [(8192, 768, 768, True), (9216, 768, 3072, True), (1000, 256, 128, True),
                                              (77, 128, 384, False), (4099, 2304, 768, True)])
def test_weight_gradient_contraction_matches_torch(M, N, K, with_bias):
    """om_gemm_tn_acc (gemm_tn.hip): C[N,K] += A[M,N]^T B[M,K], bias[N] += column sums of A on row-major bf16 operands
    -- the contraction behind every dW / db of the training backward -- against a float64 product of the same bf16
    values.  Random (asymmetric) operands, token counts that are not multiples of the 64-row step, accumulation into a
    non-zero C: any operand / output transposition or a wrong transposing-read lane mapping shows up as O(1) error."""
    from openmatch_amd import native as N_
    gen = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, N, generator=gen).to(torch.bfloat16)
    B = (torch.randn(M, K, generator=gen) * 0.5 + 0.1).to(torch.bfloat16)
    C0 = torch.randn(N, K, generator=gen)
    b0 = torch.randn(N, generator=gen)
    ref = C0.double() + A.double().t() @ B.double()
    ref_b = b0.double() + A.double().sum(0)
    Ad, Bd, Cd, bd = A.to(DEV), B.to(DEV), C0.to(DEV).contiguous(), b0.to(DEV).contiguous()
    with torch.cuda.device(DEV):
        N_.check(N_.lib().om_gemm_tn_acc(1, N_.ptr(Ad), N, N_.ptr(Bd), K, N_.ptr(Cd), K, N_.ptr(bd) if with_bias else None,
                                         M, N, K, N_.stream_ptr(Cd.device)))
    torch.cuda.synchronize()
    scale = math.sqrt(M)
    err = (Cd.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * scale * 8, (err, scale)                    # f32 accumulation of exact bf16 products
    if with_bias:
        assert (bd.cpu().double() - ref_b).abs().max().item() < 2e-5 * scale * 8
    else:
        assert torch.equal(bd.cpu(), b0)


@pytest.mark.parametrize("M", [9216, 4099, 64, 45, 20])
def test_batched_weight_gradient_contraction_matches_torch(M):
    """om_gemm_tn_acc_batch (gemm_tn.hip, gemm_tn_wide_kernel): the weight gradients of a group of layers in ONE launch --
    256 x 256 tiles over the whole token axis, plain read-add-write of C and bias.  Five problems of different shapes
    (the four of a bert-base layer and a 256 x 256 one) with non-zero C / bias, token counts that are and are not
    multiples of the 32-token step (the tail goes through the split kernels), against float64 products of the same
    bf16 values."""
    from openmatch_amd import native as N_
    shapes = [(768, 3072, True), (3072, 768, True), (768, 768, True), (2304, 768, True), (256, 256, False)]
    gen = torch.Generator().manual_seed(M)
    keep, refs = [], []
    probs = (N_.OmTnProblem * len(shapes))()
    for i, (Nn, K, with_bias) in enumerate(shapes):
        A = torch.randn(M, Nn, generator=gen).to(torch.bfloat16)
        B = (torch.randn(M, K, generator=gen) * 0.5 + 0.1).to(torch.bfloat16)
        C0 = torch.randn(Nn, K, generator=gen)
        b0 = torch.randn(Nn, generator=gen)
        Ad, Bd, Cd, bd = A.to(DEV), B.to(DEV), C0.to(DEV).contiguous(), b0.to(DEV).contiguous()
        keep.append((Ad, Bd, Cd, bd, b0))
        refs.append((C0.double() + A.double().t() @ B.double(), b0.double() + A.double().sum(0), with_bias))
        probs[i] = N_.OmTnProblem(A=N_.ptr(Ad), B=N_.ptr(Bd), C=N_.ptr(Cd), bias=N_.ptr(bd) if with_bias else None,
                                  lda=Nn, ldb=K, ldc=K, N=Nn, K=K)
    with torch.cuda.device(DEV):
        rc = N_.lib().om_gemm_tn_acc_batch(1, probs, len(shapes), M, N_.stream_ptr(keep[0][2].device))
    if M < 32:
        assert rc != 0
        return
    N_.check(rc)
    torch.cuda.synchronize()
    scale = math.sqrt(M)
    for (Ad, Bd, Cd, bd, b0), (ref, ref_b, with_bias) in zip(keep, refs):
        err = (Cd.cpu().double() - ref).abs().max().item()
        assert err < 2e-5 * scale * 8, (err, scale, tuple(ref.shape))
        if with_bias:
            assert (bd.cpu().double() - ref_b).abs().max().item() < 2e-5 * scale * 8
        else:
            assert torch.equal(bd.cpu(), b0)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_deferred_batched_weight_gradients_equal_per_site_launches(p_drop):
    """OM_OPT_TRAIN_WGRAD_BATCH: the bf16 BERT backward keeps every layer's dY and computes the weight gradients of a
    group of layers in one om_gemm_tn_acc_batch launch (train.hip).  Same step -- three bert-base-width layers, 6 x 128 +
    3 x 32 tokens, same dropout seed -- with per-site launches (0), groups of two layers (the last group is a single
    layer) and one group for the whole stack, each with and without the side stream: every parameter gradient agrees to
    f32 accumulation-order noise (the data-gradient chain is the same kernels in all of them)."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch_amd import native as N_
    torch.manual_seed(21)
    cfg = BertConfig(hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072,
                     vocab_size=600, max_position_embeddings=160, hidden_dropout_prob=p_drop,
                     attention_probs_dropout_prob=p_drop)
    lm = BertModel(cfg)
    rng = np.random.default_rng(9)
    p_ids, p_mask = synth_tokens(rng, 6, 128, vocab=600, lo_len=16, lo_id=300)
    q_ids, q_mask = synth_tokens(rng, 3, 32, vocab=600, lo_len=4, lo_id=300)
    tens = lambda a: torch.from_numpy(a).to(DEV)
    model = DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="bfloat16"),
                    data_args=NS(train_n_passages=2),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=3)).to(DEV).train()
    runs = {}
    try:
        for batch, lane in ((0, 1), (2, 1), (2, 0), (12, 1), (12, 0)):
            N_.check(N_.lib().om_debug_option(14, batch))
            N_.check(N_.lib().om_debug_option(8, lane))
            model.zero_grad(set_to_none=True)
            torch.manual_seed(77)                                    # the step's dropout seed
            out = model(query={"input_ids": tens(q_ids), "attention_mask": tens(q_mask)},
                        passage={"input_ids": tens(p_ids), "attention_mask": tens(p_mask)})
            out.loss.backward()
            torch.cuda.synchronize()
            runs[(batch, lane)] = (out.loss.item(), {n: p.grad.detach().float().cpu().clone()
                                                    for n, p in lm.named_parameters() if p.grad is not None})
    finally:
        N_.check(N_.lib().om_debug_option(14, 4))
        N_.check(N_.lib().om_debug_option(8, 1))
    l0, g0 = runs[(0, 1)]
    assert any("query.weight" in n for n in g0) and any("output.dense.weight" in n for n in g0)
    for key, (l1, g1) in runs.items():
        assert abs(l1 - l0) < 1e-6 * max(1.0, abs(l0)), (key, l0, l1)
        assert g1.keys() == g0.keys()
        for n in g0:
            rel = ((g1[n] - g0[n]).norm() / (g0[n].norm() + 1e-20)).item()
            assert rel < 1e-4 or (g1[n] - g0[n]).abs().max().item() < 1e-7, (key, n, rel)


def test_tied_training_forward_in_one_pass_equals_two_calls(golden):
    """A tied DRModel in training mode pads the queries to the passage length and encodes both batches in one pass
    (modeling/dense_retrieval_model.py: _encode_one_pass); the reference calls the shared module twice (:89-93).  Same
    loss, scores and parameter gradients as the two-call path (taken in eval mode: dropout is 0 in both)."""
    g = golden("train_bert_tiny")
    q, p = _train_batch(g)
    p2 = {k: torch.cat([v, v.flip(0)], 0) for k, v in p.items()}          # 4 x as many passages as queries: one pass
    results = []
    for one_pass in (True, False):
        model = _train_model(g)
        model.data_args = NS(train_n_passages=2 * int(g["n_psg"]))
        if not one_pass:
            model.eval()
        assert model._one_pass_ok(q, p2) == one_pass
        out = model(query=q, passage=p2)
        out.loss.backward()
        results.append((out.loss.item(), out.scores.detach().cpu(),
                        {n: t.grad.detach().cpu().clone() for n, t in model.lm_q.named_parameters() if t.grad is not None},
                        model.head_q.linear.weight.grad.detach().cpu().clone()))
    (l1, s1, g1, h1), (l2, s2, g2, h2) = results
    assert abs(l1 - l2) < 1e-5 and (s1 - s2).abs().max().item() < 1e-5
    assert g1.keys() == g2.keys()
    for n in g1:
        rel = ((g1[n] - g2[n]).norm() / (g2[n].norm() + 1e-20)).item()
        assert rel < 1e-4 or (g1[n] - g2[n]).abs().max().item() < 1e-7, (n, rel)
    assert ((h1 - h2).norm() / h2.norm()).item() < 1e-4


@pytest.mark.parametrize("L,p_drop", [(128, 0.0), (128, 0.1), (80, 0.1), (40, 0.1), (20, 0.0)])
def test_bf16_attention_backward_kernels_agree(L, p_drop):
    """The bf16 training backward has two attention kernels: the transposing-read one (attention_bwd16.hip, L <= 128) and
    the generic one (train_kernels.hip; OM_OPT_ATTENTION_FAST = 0 selects it).  Same model, batch and dropout seed ->
    the same masks (both regenerate them from the seed) and the same gradients up to bf16 rounding of intermediates,
    for full and partially filled 32-token tiles and ragged attention masks."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch_amd import native as N_
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                     max_position_embeddings=128, hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop)
    torch.manual_seed(21)
    lm = BertModel(cfg)
    rng = np.random.default_rng(L)
    p_ids, p_mask = synth_tokens(rng, 12, L, vocab=600, lo_len=max(3, L // 3), lo_id=300)
    q_ids, q_mask = synth_tokens(rng, 3, L, vocab=600, lo_len=3, lo_id=300)
    tens = lambda a: torch.from_numpy(a).to(DEV)
    grads = []
    for fast in (1, 0):
        model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=False, dtype="bfloat16"),
                        data_args=NS(train_n_passages=4),
                        train_args=NS(negatives_x_device=False, per_device_train_batch_size=3)).to(DEV).train()
        model.zero_grad(set_to_none=True)
        torch.manual_seed(77)                                # the dropout seed of the step is drawn from torch's generator
        N_.check(N_.lib().om_debug_option(2, fast))
        try:
            out = model(query={"input_ids": tens(q_ids), "attention_mask": tens(q_mask)},
                        passage={"input_ids": tens(p_ids), "attention_mask": tens(p_mask)})
            out.loss.backward()
        finally:
            N_.check(N_.lib().om_debug_option(2, 1))
        grads.append((out.loss.item(), {n: t.grad.detach().float().cpu().clone() for n, t in lm.named_parameters() if t.grad is not None}))
    (l1, g1), (l2, g2) = grads
    assert abs(l1 - l2) < 1e-3 * max(1.0, abs(l2))           # (without dropout the switch also changes the forward attention kernel)
    worst = ("", 1.0)
    for n in g2:
        a, b_ = g1[n].flatten().double(), g2[n].flatten().double()
        if b_.norm() < 1e-9 or n.endswith("attention.self.key.bias"):
            continue      # (d loss / d key bias is identically zero -- softmax ignores a per-query shift of the scores: noise only)
        cos = (torch.dot(a, b_) / (a.norm() * b_.norm())).item()
        rel = ((a - b_).norm() / b_.norm()).item()
        worst = min(worst, (n, cos), key=lambda t: t[1])
        assert cos > 0.999 and rel < 3e-2, (L, p_drop, n, cos, rel)
    print(f"\n[attention backward, L={L}, p={p_drop}] least aligned gradient tensor: {worst}")


@pytest.mark.parametrize("arch", ["bert", "t5"])
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_tile_at_a_time_attention_kernels_agree_with_the_others_under_dropout(dtype, arch, monkeypatch):
    """Training beyond 256 tokens (round 6) runs its attention on kernels that keep ONE score tile in registers: the online-softmax
    forward (with dropout now) and attention_bwd_long_a / _b_kernel (two passes over the key tiles, delta = dO . O; dK, dV over the query tiles).  Forced at L = 200 / 96
    (OM_OPT_ATTENTION_FAST bit 1) they must reproduce the step of the kernels that normally serve those lengths: the same dropout masks
    (one hash of (sequence, head, query, key)), loss and gradients up to 16-bit rounding -- BERT and T5 (position bias and its gradient)."""
    monkeypatch.delenv("OM_TRAIN_F16", raising=False); monkeypatch.delenv("OM_T5_F16", raising=False)
    from transformers import BertConfig, BertModel, T5Config, T5EncoderModel
    from openmatch.modeling import DRModel
    from openmatch_amd import native as N_
    torch.manual_seed(23)
    for L, p_drop in ((200, 0.0), (200, 0.1), (96, 0.1)):
        if arch == "bert":
            lm = BertModel(BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                                      max_position_embeddings=256, hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop))
        else:
            lm = T5EncoderModel(T5Config(d_model=128, d_ff=256, num_layers=2, num_heads=2, d_kv=64, vocab_size=600, feed_forward_proj="relu", dropout_rate=p_drop))
        rng = np.random.default_rng(L)
        p_ids, p_mask = synth_tokens(rng, 8, L, vocab=600, lo_len=max(3, L // 3), lo_id=300)
        q_ids, q_mask = synth_tokens(rng, 2, L, vocab=600, lo_len=3, lo_id=300)
        tens = lambda a: torch.from_numpy(a).to(DEV)
        grads = []
        for opt in (1, 3):
            model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=arch == "t5", dtype=dtype),
                            data_args=NS(train_n_passages=4),
                            train_args=NS(negatives_x_device=False, per_device_train_batch_size=2)).to(DEV).train()
            model.zero_grad(set_to_none=True)
            torch.manual_seed(77)
            N_.check(N_.lib().om_debug_option(2, opt))
            try:
                out = model(query={"input_ids": tens(q_ids), "attention_mask": tens(q_mask)},
                            passage={"input_ids": tens(p_ids), "attention_mask": tens(p_mask)})
                out.loss.backward()
            finally:
                N_.check(N_.lib().om_debug_option(2, 1))
            grads.append((out.loss.item(), {n: t.grad.detach().float().cpu().clone() for n, t in lm.named_parameters() if t.grad is not None}))
        (l1, g1), (l2, g2) = grads
        assert abs(l1 - l2) < 2e-3 * max(1.0, abs(l1)), (L, p_drop, l1, l2)
        worst = ("", 1.0, 0.0)
        for n in g1:
            a, b_ = g2[n].flatten().double(), g1[n].flatten().double()
            if b_.norm() < 1e-9 or n.endswith("attention.self.key.bias"):
                continue
            cos = (torch.dot(a, b_) / (a.norm() * b_.norm())).item()
            rel = ((a - b_).norm() / b_.norm()).item()
            if cos < worst[1]:
                worst = (n, cos, rel)
            assert cos > (0.999 if arch == "bert" else 0.998) and rel < (3e-2 if arch == "bert" else 6e-2), (arch, dtype, L, p_drop, n, cos, rel)      # (T5: ReLU patterns flip between two 16-bit forwards)
        print(f"\n[tile-at-a-time attention, {arch}, {dtype}, L={L}, p={p_drop}] loss {l2:.5f} vs {l1:.5f}; least aligned gradient tensor: {worst}")


@pytest.mark.parametrize("arch", ["bert", "t5"])
@pytest.mark.parametrize("L", [320, 512])
def test_training_step_beyond_256_tokens_matches_torch_autograd(L, arch, monkeypatch):
    """Round 6 (VERDICT r5 "missing" 3): training at 257 .. 512 tokens in the 16-bit formats.  Loss and every parameter gradient of a
    contrastive step (no dropout) against torch autograd through the HF module in fp32, on ragged right-padded batches; and with dropout
    the step stays finite and moves."""
    monkeypatch.delenv("OM_TRAIN_F16", raising=False); monkeypatch.delenv("OM_T5_F16", raising=False)
    from transformers import BertConfig, BertModel, T5Config, T5EncoderModel
    from openmatch.modeling import DRModel
    torch.manual_seed(29)
    if arch == "bert":
        mk = lambda: BertModel(BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                                          max_position_embeddings=512, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    else:
        mk = lambda: T5EncoderModel(T5Config(d_model=128, d_ff=256, num_layers=2, num_heads=2, d_kv=64, vocab_size=600, feed_forward_proj="relu", dropout_rate=0.0))
    lm = mk()
    ref_lm = mk(); ref_lm.load_state_dict(lm.state_dict())
    rng = np.random.default_rng(L + 1)
    p_ids, p_mask = synth_tokens(rng, 6, L, vocab=600, lo_len=L // 2, lo_id=300)
    q_ids, q_mask = synth_tokens(rng, 2, L, vocab=600, lo_len=5, lo_id=300)
    p_ids[0, :], p_mask[0, :] = rng.integers(300, 600, L), 1          # a full-length row
    margs = lambda dt: NS(encoder_only=arch == "t5", dtype=dt)
    common = dict(data_args=NS(train_n_passages=3), train_args=NS(negatives_x_device=False, per_device_train_batch_size=2))
    def ref_mean(ids, mask):      # the reference's encode (mean pooling) through the HF module, fp32 on the CPU
        ids, mask = torch.from_numpy(ids), torch.from_numpy(mask)
        h_ = ref_lm(input_ids=ids, attention_mask=mask).last_hidden_state
        m_ = mask.unsqueeze(-1).float()
        return (h_ * m_).sum(1) / m_.sum(1).clamp(min=1e-9)
    ref_lm.train()
    loss_ref, _ = retrieval_ref.contrastive_loss(ref_mean(q_ids, q_mask), ref_mean(p_ids, p_mask), 3)
    loss_ref.backward()
    ref_out = NS(loss=loss_ref.detach())
    gref = {n: t.grad.detach().clone() for n, t in ref_lm.named_parameters() if t.grad is not None}
    tens = lambda a: torch.from_numpy(a).to(DEV)
    for dtype in ("float16", "bfloat16"):
        model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", model_args=margs(dtype), **common).to(DEV).train()
        model.zero_grad(set_to_none=True)
        out = model(query={"input_ids": tens(q_ids), "attention_mask": tens(q_mask)}, passage={"input_ids": tens(p_ids), "attention_mask": tens(p_mask)})
        lscale = 4096.0 if dtype == "float16" else 1.0       # float16 trains under a loss scale (the reference's GradScaler; DRTrainer's device-side one): a mean over
        (out.loss * lscale).backward()                        # 512 tokens puts unscaled score gradients below float16's smallest normal
        for t in lm.parameters():
            if t.grad is not None:
                t.grad /= lscale
        tol = 2e-3 if dtype == "float16" else 2e-2
        assert abs(out.loss.item() - ref_out.loss.item()) < tol * max(1.0, abs(ref_out.loss.item())), (dtype, out.loss.item(), ref_out.loss.item())
        worst = ("", 0.0)
        for n, t in lm.named_parameters():
            if n not in gref or gref[n].norm() < 1e-9 or n.endswith("attention.self.key.bias"):
                continue
            rel = ((t.grad.detach().float().cpu() - gref[n]).norm() / gref[n].norm()).item()
            if rel > worst[1]:
                worst = (n, rel)
        print(f"\n[training at L={L}, {arch}, {dtype}] loss {out.loss.item():.5f} vs torch fp32 {ref_out.loss.item():.5f}; worst gradient rel-L2 {worst[1]:.2e} ({worst[0]})")
        assert worst[1] < (3e-2 if dtype == "float16" else 8e-2), (dtype, worst)      # (16-bit steps against fp32 autograd on a tiny random-init model: T5's ReLU pattern flips at ~1e-2)
    if arch == "bert":
        lm.config.hidden_dropout_prob = lm.config.attention_probs_dropout_prob = 0.1
    else:
        lm.config.dropout_rate = 0.1
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", model_args=margs("float16"), **common).to(DEV).train()
    model.zero_grad(set_to_none=True)
    out_d = model(query={"input_ids": tens(q_ids), "attention_mask": tens(q_mask)}, passage={"input_ids": tens(p_ids), "attention_mask": tens(p_mask)})
    out_d.loss.backward()
    assert math.isfinite(out_d.loss.item()) and out_d.loss.item() != out.loss.item()
    assert all(torch.isfinite(t.grad).all() for t in lm.parameters() if t.grad is not None)


def test_roberta_backbone_matches_hf_forward_and_autograd():
    """`AutoModel.from_pretrained` (reference modeling/dense_retrieval_model.py:173) may hand DRModel a RoBERTa-family
    checkpoint: the BERT stack behind position ids that start at padding_idx + 1 (HF create_position_ids_from_input_ids).
    f32 HIP path vs HF RobertaModel on the CPU, right-padded ragged batches: embeddings within 1e-4, and loss + every
    parameter gradient of a contrastive training step vs torch autograd through the HF module (incl. the shifted rows
    of the position table)."""
    from transformers import RobertaConfig, RobertaModel
    from openmatch.modeling import DRModel, DRModelForInference
    torch.manual_seed(31)
    cfg = RobertaConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                        max_position_embeddings=140, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = RobertaModel(cfg)
    ref_lm = RobertaModel(cfg); ref_lm.load_state_dict(lm.state_dict())
    pad = cfg.pad_token_id
    rng = np.random.default_rng(3)

    def batch(n, L, lo):
        ids, mask = synth_tokens(rng, n, L, vocab=600, lo_len=lo, lo_id=300)
        ids[mask == 0] = pad                                  # HF derives the positions from input_ids != pad
        return torch.from_numpy(ids), torch.from_numpy(mask)
    p_ids, p_mask = batch(6, 128, 40)
    q_ids, q_mask = batch(3, 32, 5)

    def ref_mean(ids, mask):
        h = ref_lm(input_ids=ids, attention_mask=mask).last_hidden_state
        m = mask.unsqueeze(-1).float()
        return (h * m).sum(1) / m.sum(1).clamp(min=1e-9)
    # inference
    inf = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=False, dtype="float32")).to(DEV).eval()
    with torch.no_grad():
        want = ref_mean(p_ids, p_mask)
        _, got = inf.encode_passage({"input_ids": p_ids.to(DEV), "attention_mask": p_mask.to(DEV)})
    assert (got.cpu() - want).abs().max().item() < 1e-4
    # training step
    hq, hp = ref_mean(q_ids, q_mask), ref_mean(p_ids, p_mask)
    loss_ref, _ = retrieval_ref.contrastive_loss(hq, hp, 2)
    loss_ref.backward()
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=False, dtype="float32"),
                    data_args=NS(train_n_passages=2),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=3)).to(DEV).train()
    out = model(query={"input_ids": q_ids.to(DEV), "attention_mask": q_mask.to(DEV)},
                passage={"input_ids": p_ids.to(DEV), "attention_mask": p_mask.to(DEV)})
    assert abs(out.loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item()))
    out.loss.backward()
    got_g = dict(lm.named_parameters())
    checked = 0
    for name, p in ref_lm.named_parameters():
        if p.grad is None or "pooler" in name:
            continue
        gg = got_g[name].grad.cpu()
        if name == "embeddings.position_embeddings.weight":   # rows of padded positions: HF adds them to row `pad`, masked here
            rows = slice(pad + 1, pad + 1 + 128)
            gg, ref_g = gg[rows], p.grad[rows]
        elif name == "embeddings.word_embeddings.weight":
            keep = torch.ones(cfg.vocab_size, dtype=torch.bool); keep[pad] = False
            gg, ref_g = gg[keep], p.grad[keep]
        else:
            ref_g = p.grad
        rel = ((gg - ref_g).norm() / (ref_g.norm() + 1e-20)).item()
        assert rel < 2e-3 or (gg - ref_g).abs().max().item() < 1e-7, (name, rel)
        checked += 1
    assert checked >= 30


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_t5_decoder_position_over_long_passages_matches_hf(dtype, monkeypatch):
    """Round 6: the decoder position (monoT5 scoring, encoder-decoder pooling) over encoder outputs of up to 1 024 tokens -- monoT5 re-rankers
    run at 512 (the cross-attention kernels were built for <= 256 encoder positions) -- and its TRAINING at up to 512.  Inference at 320 and
    700 tokens against HF T5ForConditionalGeneration in f32 on the CPU; a training step at 320 tokens against torch autograd (no dropout)."""
    monkeypatch.delenv("OM_TRAIN_F16", raising=False); monkeypatch.delenv("OM_T5_F16", raising=False)
    from transformers import T5Config, T5ForConditionalGeneration
    from openmatch.modeling import DRModel, DRModelForInference
    torch.manual_seed(47)
    cfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, d_kv=64, vocab_size=600,
                   feed_forward_proj="relu", decoder_start_token_id=0, dropout_rate=0.0)
    lm = T5ForConditionalGeneration(cfg).eval()
    cpu_lm = T5ForConditionalGeneration(cfg).eval(); cpu_lm.load_state_dict(lm.state_dict())      # (the HIP wrappers move `lm` to the device)
    rng = np.random.default_rng(11)
    for L in (320, 700):
        ids, mask = synth_tokens(rng, 5, L, vocab=600, lo_len=L // 2, lo_id=300)
        ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
        with torch.no_grad():
            out = cpu_lm(input_ids=ids_t, attention_mask=mask_t, decoder_input_ids=torch.zeros(5, 1, dtype=torch.long), output_hidden_states=True, return_dict=True)
            want = out.decoder_hidden_states[-1][:, 0, :]
        dr = DRModelForInference(lm_q=lm, lm_p=lm, model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
        hidden, _ = dr.encode_passage({"input_ids": ids_t.to(DEV), "attention_mask": mask_t.to(DEV)})
        cos = torch.nn.functional.cosine_similarity(hidden[:, 0, :].float().cpu(), want, dim=1).min().item()
        assert cos > (0.9999 if dtype == "float16" else 0.999), (L, cos)
    # training at 320 tokens: the decoder state as the representation, contrastive loss, against torch autograd through the HF module
    L = 320
    ref_lm = T5ForConditionalGeneration(cfg); ref_lm.load_state_dict(lm.state_dict()); ref_lm.train()
    p_ids, p_mask = synth_tokens(rng, 4, L, vocab=600, lo_len=L // 2, lo_id=300)
    q_ids, q_mask = synth_tokens(rng, 2, L, vocab=600, lo_len=5, lo_id=300)
    def ref_state(ids, mask):
        o = ref_lm(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), decoder_input_ids=torch.zeros(len(ids), 1, dtype=torch.long),
                   output_hidden_states=True, return_dict=True)
        return o.decoder_hidden_states[-1][:, 0, :]
    loss_ref, _ = retrieval_ref.contrastive_loss(ref_state(q_ids, q_mask), ref_state(p_ids, p_mask), 2)
    loss_ref.backward()
    model = DRModel(lm_q=lm, lm_p=lm, model_args=NS(encoder_only=False, dtype=dtype), data_args=NS(train_n_passages=2),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=2)).to(DEV).train()
    model.zero_grad(set_to_none=True)
    tens = lambda a: torch.from_numpy(a).to(DEV)
    out = model(query={"input_ids": tens(q_ids), "attention_mask": tens(q_mask)}, passage={"input_ids": tens(p_ids), "attention_mask": tens(p_mask)})
    lscale = 1024.0 if dtype == "float16" else 1.0
    (out.loss * lscale).backward()
    assert abs(out.loss.item() - loss_ref.item()) < (5e-3 if dtype == "float16" else 3e-2) * max(1.0, abs(loss_ref.item())), (out.loss.item(), loss_ref.item())
    worst = ("", 0.0)
    gref = {n: t.grad for n, t in ref_lm.named_parameters() if t.grad is not None}
    for n, t in lm.named_parameters():
        if t.grad is None or n not in gref or gref[n].norm() < 1e-9:
            continue
        rel = ((t.grad.detach().float().cpu() / lscale - gref[n]).norm() / gref[n].norm()).item()
        if rel > worst[1]:
            worst = (n, rel)
    print(f"\n[T5 decoder position, training at L={L}, {dtype}] loss {out.loss.item():.5f} vs torch fp32 {loss_ref.item():.5f}; worst gradient rel-L2 {worst[1]:.2e} ({worst[0]})")
    # float16: every tensor; bfloat16: single LayerNorm gains are sums that cancel to 16-bit noise on this tiny model (0.33 on one of them) --
    # the WHOLE gradient is held instead
    got_all = torch.cat([(t.grad.detach().float().cpu() / lscale).flatten() for n, t in lm.named_parameters() if t.grad is not None and n in gref])
    ref_all = torch.cat([gref[n].flatten() for n, t in lm.named_parameters() if t.grad is not None and n in gref])
    whole = ((got_all - ref_all).norm() / ref_all.norm()).item()
    assert whole < (1e-2 if dtype == "float16" else 1e-1), whole      # (bfloat16: 7e-2 measured; float16, the same arithmetic with three more mantissa bits, is at 1e-2 per tensor)
    if dtype == "float16":
        assert worst[1] < 5e-2, worst


@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_t5_encoder_decoder_pooling_and_monot5_match_hf(gated, dtype):
    """T5 backbones that are NOT --encoder_only: the reference feeds decoder_input_ids = zeros([B, 1]) and takes the
    decoder's hidden state as the representation (modeling/dense_retrieval_model.py:137-141), or -- monoT5 -- the LM
    head's logits of two tokens there (modeling/reranking_model.py:110-114, + log_softmax in retriever/reranker.py:115).
    HIP path (om_encoder_forward + om_t5_decoder_step) vs HF T5ForConditionalGeneration on the CPU in f32."""
    from transformers import T5Config, T5ForConditionalGeneration
    from openmatch.modeling import DRModelForInference, LinearHead, RRModel
    torch.manual_seed(41 + gated)
    cfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_decoder_layers=3, num_heads=2, d_kv=64, vocab_size=600,
                   feed_forward_proj="gated-gelu" if gated else "relu", tie_word_embeddings=not gated,
                   decoder_start_token_id=0)
    lm = T5ForConditionalGeneration(cfg).eval()
    with torch.no_grad():                                   # random init leaves the norms at 1: make them matter
        for name, p in lm.named_parameters():
            if "layer_norm" in name:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
    rng = np.random.default_rng(9)
    ids, mask = synth_tokens(rng, 7, 96, vocab=600, lo_len=20, lo_id=300)
    ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
    with torch.no_grad():
        out = lm(input_ids=ids_t, attention_mask=mask_t, decoder_input_ids=torch.zeros(7, 1, dtype=torch.long),
                 output_hidden_states=True, return_dict=True)
        want_state = out.decoder_hidden_states[-1][:, 0, :]             # after the decoder's final norm
        want_logits = out.logits[:, 0, [17, 23]]
    items = {"input_ids": ids_t.to(DEV), "attention_mask": mask_t.to(DEV)}
    exact = dtype == "float32"
    # --- bi-encoder pooling: DRModel.encode -> (hidden [B,1,H], reps)
    head = LinearHead(128, 64)
    dr = DRModelForInference(lm_q=lm, lm_p=lm, head_q=head, head_p=head, normalize=True,
                             model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
    hidden, reps = dr.encode_passage(items)
    assert hidden.shape == (7, 1, 128) and reps.shape == (7, 64)
    got_state = hidden[:, 0, :].float().cpu()
    want_reps = torch.nn.functional.normalize(want_state @ head.linear.weight.detach().cpu().t(), dim=1)
    if exact:
        assert (got_state - want_state).abs().max().item() < 1e-4 * max(1.0, want_state.abs().max().item())
        assert (reps.cpu() - want_reps).abs().max().item() < 1e-4
    else:
        cos = torch.nn.functional.cosine_similarity(got_state, want_state, dim=1).min().item()
        assert cos > 0.999, cos
    # --- monoT5: RRModel.encode -> [B, 2] logits of (neg_token, pos_token)
    class Tok:                                               # RRModel only asks the tokenizer for the two token ids
        def encode(self, t, add_special_tokens=False):
            return [{"true": 23, "false": 17}[t]]
    rr = RRModel(lm=lm, head=LinearHead(128, 1), pos_token="true", neg_token="false", tokenizer=Tok(),
                 model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
    with torch.no_grad():
        got_logits = rr.encode(items).cpu()
    assert got_logits.shape == (7, 2)
    if exact:
        assert (got_logits - want_logits).abs().max().item() < 1e-4 * max(1.0, want_logits.abs().max().item())
    else:
        want_s = torch.log_softmax(want_logits, dim=1)[:, 1]
        got_s = torch.log_softmax(got_logits, dim=1)[:, 1]
        assert (got_s - want_s).abs().max().item() < 4e-2 * max(1.0, want_logits.abs().max().item())      # bf16: relative to the logits' size (measured 3.05e-2)


@pytest.mark.parametrize("gated", [False, True])
def test_t5_encoder_decoder_training_matches_hf_autograd(gated):
    """Training through the decoder position (reference: autograd under DRModel.encode :137-141 and RRModel.encode
    :110-114): f32 HIP path (om_encoder_train_*_hidden + om_t5_decoder_train_*) vs HF T5ForConditionalGeneration autograd
    on the CPU, dropout 0 -- the representations / logits and EVERY parameter gradient: shared embedding (encoder tokens
    + the decoder's start row [+ the tied LM-head rows]), both stacks' q/k/v/o, RMSNorm weights, feed-forward weights,
    the encoder's relative-position table; the decoder's self-attention q / k and its position table take part in HF's
    graph with a zero gradient (softmax over one key) and must come out as zeros, not None."""
    import copy
    from transformers import T5Config, T5ForConditionalGeneration
    from openmatch.modeling import DRModel, LinearHead, RRModel
    torch.manual_seed(77 + gated)
    cfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_decoder_layers=3, num_heads=2, d_kv=64, vocab_size=600,
                   feed_forward_proj="gated-gelu" if gated else "relu", tie_word_embeddings=not gated,
                   decoder_start_token_id=0, dropout_rate=0.0)
    lm = T5ForConditionalGeneration(cfg)
    with torch.no_grad():
        for name, p in lm.named_parameters():
            if "layer_norm" in name:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))
            elif "relative_attention_bias" in name:
                p.copy_(0.5 * torch.randn_like(p))
    rng = np.random.default_rng(19)
    ids, mask = synth_tokens(rng, 6, 80, vocab=600, lo_len=12, lo_id=300)
    ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
    items = {"input_ids": ids_t.to(DEV), "attention_mask": mask_t.to(DEV)}
    head = LinearHead(128, 64)
    R = torch.randn(6, 64, generator=torch.Generator().manual_seed(5))
    R2 = torch.randn(6, 2, generator=torch.Generator().manual_seed(6))

    def hf_state(ref):
        enc = ref.encoder(input_ids=ids_t, attention_mask=mask_t).last_hidden_state
        return ref.decoder(input_ids=torch.zeros(6, 1, dtype=torch.long), encoder_hidden_states=enc,
                           encoder_attention_mask=mask_t).last_hidden_state[:, 0]

    def compare(model_lm, ref, extra=()):
        names = dict(model_lm.named_parameters())
        checked = zeros = 0
        seen = set()
        for name, rp in ref.named_parameters():
            if id(rp) in seen:
                continue
            seen.add(id(rp))
            got = names[name].grad
            if rp.grad is None:
                assert got is None or float(got.abs().max()) == 0.0, name
                continue
            assert got is not None, name
            got, want = got.float().cpu(), rp.grad
            if float(want.abs().max()) == 0.0:
                assert float(got.abs().max()) == 0.0, name
                zeros += 1
                continue
            rel = ((got - want).norm() / (want.norm() + 1e-20)).item()
            amax = (got - want).abs().max().item()
            assert rel < 2e-3 or amax < 1e-7, (name, rel, amax)
            checked += 1
        for got, want, name in extra:
            rel = ((got.float().cpu() - want).norm() / (want.norm() + 1e-20)).item()
            assert rel < 2e-3, (name, rel)
        return checked, zeros

    # --- bi-encoder pooling through the decoder position, head + normalise
    ref = copy.deepcopy(lm)
    ref_head_w = head.linear.weight.detach().clone().requires_grad_()          # (LinearHead itself is HIP-only: plain matmul here)
    want_reps = torch.nn.functional.normalize(hf_state(ref) @ ref_head_w.t(), dim=1)
    (want_reps * R).sum().backward()
    dr = DRModel(lm_q=lm, lm_p=lm, head_q=head, head_p=head, normalize=True, model_args=NS(encoder_only=False, dtype="float32"),
                 data_args=NS(train_n_passages=1), train_args=NS(negatives_x_device=False)).to(DEV).train()
    hidden, reps = dr.encode_passage(items)
    assert hidden.shape == (6, 1, 128) and reps.requires_grad
    assert (reps.detach().cpu() - want_reps.detach()).abs().max().item() < 1e-4
    (reps * R.to(DEV)).sum().backward()
    checked, zeros = compare(dr.lm_p, ref, extra=[(head.linear.weight.grad, ref_head_w.grad, "head")])
    n_dec, n_enc = 3, 2
    assert zeros == 2 * n_dec + 1, zeros                      # decoder self-attention q, k per block + its position table
    assert checked >= 1 + n_enc * (8 if gated else 7) + 2 + n_dec * (11 if gated else 10) + 1, checked
    # --- monoT5: two LM-head rows of the same state
    for p_ in list(dr.lm_p.parameters()) + list(head.parameters()):
        p_.grad = None
    ref2 = copy.deepcopy(lm).cpu()
    want_logits = ref2(input_ids=ids_t, attention_mask=mask_t, decoder_input_ids=torch.zeros(6, 1, dtype=torch.long),
                       return_dict=True).logits[:, 0, [17, 23]]
    (want_logits * R2).sum().backward()

    class Tok:
        def encode(self, t, add_special_tokens=False):
            return [{"true": 23, "false": 17}[t]]
    rr = RRModel(lm=dr.lm_p, head=LinearHead(128, 1), pos_token="true", neg_token="false", tokenizer=Tok(),
                 model_args=NS(encoder_only=False, dtype="float32")).to(DEV).train()
    logits = rr.encode(items)
    assert logits.shape == (6, 2) and logits.requires_grad
    assert (logits.detach().cpu() - want_logits.detach()).abs().max().item() < 1e-4 * max(1.0, want_logits.abs().max().item())
    (logits * R2.to(DEV)).sum().backward()
    checked2, zeros2 = compare(rr.lm, ref2)
    assert zeros2 == 2 * n_dec + 1 and checked2 >= checked, (checked2, zeros2)


def test_t5_encoder_decoder_training_bf16_and_dropout_are_sane():
    """bf16 gradients through the decoder position track the f32 ones; with dropout_rate 0.1 the step stays finite, masks
    change from call to call, eval() is deterministic, and a step along -grad lowers the (eval) loss."""
    import copy
    from transformers import T5Config, T5ForConditionalGeneration
    from openmatch.modeling import DRModel
    torch.manual_seed(123)
    cfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, d_kv=64, vocab_size=600,
                   feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0, dropout_rate=0.0)
    lm = T5ForConditionalGeneration(cfg)
    rng = np.random.default_rng(23)
    ids, mask = synth_tokens(rng, 8, 64, vocab=600, lo_len=10, lo_id=300)
    items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    R = torch.randn(8, 128, generator=torch.Generator().manual_seed(8)).to(DEV)

    def grads(dtype):
        m = copy.deepcopy(lm)
        dr = DRModel(lm_q=m, lm_p=m, model_args=NS(encoder_only=False, dtype=dtype), data_args=NS(train_n_passages=1),
                     train_args=NS(negatives_x_device=False)).to(DEV).train()
        _h, reps = dr.encode_passage(items)
        (reps * R).sum().backward()
        return {n: p.grad.float().cpu() for n, p in m.named_parameters() if p.grad is not None}
    g32, g16 = grads("float32"), grads("bfloat16")
    for key in ("shared.weight", "encoder.block.1.layer.0.SelfAttention.o.weight", "decoder.block.0.layer.1.EncDecAttention.k.weight",
                "decoder.block.1.layer.2.DenseReluDense.wi_1.weight", "decoder.block.0.layer.0.SelfAttention.v.weight",
                "decoder.final_layer_norm.weight"):
        a, b = g32[key].flatten().double(), g16[key].flatten().double()
        cos = torch.dot(a, b) / (a.norm() * b.norm())
        assert cos > 0.99, (key, cos.item())
    m = copy.deepcopy(lm)
    m.config.dropout_rate = 0.1
    dr = DRModel(lm_q=m, lm_p=m, model_args=NS(encoder_only=False, dtype="float32"), data_args=NS(train_n_passages=1),
                 train_args=NS(negatives_x_device=False)).to(DEV).train()
    loss = lambda: (dr.encode_passage(items)[1] * R).sum()
    l1, l2 = loss(), loss()
    assert torch.isfinite(l1) and torch.isfinite(l2) and l1.item() != l2.item()
    l1.backward()
    params = [p_ for p_ in m.parameters() if p_.grad is not None]
    assert all(torch.isfinite(p_.grad).all() for p_ in params)
    dr.eval()
    with torch.no_grad():
        a, b = loss().item(), loss().item()
        assert a == b
    # the dropout-free gradient is a descent direction of the dropout-free loss
    m.config.dropout_rate = 0.0
    dr.train()
    for p_ in params:
        p_.grad = None
    before = loss()
    before.backward()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.grad is not None:
                p_.add_(p_.grad, alpha=-0.02)
        after = loss().item()
    assert after < before.item(), (before.item(), after)


@pytest.mark.parametrize("arch,L,dtype", [("bert", 384, "float32"), ("bert", 512, "float32"), ("bert", 512, "bfloat16"),
                                          ("bert", 320, "float16"), ("bert", 512, "float16"),
                                          ("t5", 320, "float32"), ("t5", 512, "bfloat16")])
def test_long_sequences_match_oracle(arch, L, dtype):
    """Document-length inputs (the reference accepts anything up to max_position_embeddings; its document recipes use
    512): above 256 tokens attention runs on the key-chunked online-softmax kernel (attention.hip:
    attention_long_kernel).  Ragged batches whose lengths straddle the 128-key chunks, against the CPU oracle."""
    from openmatch.modeling import DRModelForInference
    torch.manual_seed(L)
    if arch == "bert":
        from transformers import BertConfig, BertModel
        cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                         max_position_embeddings=512)
        lm = BertModel(cfg).eval()
        encoder_only = False
    else:
        from transformers import T5Config, T5EncoderModel
        cfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_heads=2, d_kv=64, vocab_size=600)
        lm = T5EncoderModel(cfg).eval()
        with torch.no_grad():
            for name, p in lm.named_parameters():
                if "relative_attention_bias" in name:
                    p.copy_(0.5 * torch.randn_like(p))
        encoder_only = True
    rng = np.random.default_rng(L)
    ids, mask = synth_tokens(rng, 5, L, vocab=600, lo_len=L // 4, lo_id=300)
    ids[0, :], mask[0, :] = rng.integers(300, 600, L), 1          # one full-length row
    tens = lambda a: torch.from_numpy(a)
    sd = dict(lm.named_parameters()); sd.update({k: v for k, v in lm.named_buffers()})
    with torch.no_grad():
        ref = encoder_ref.encode(sd, cfg, arch, {"input_ids": tens(ids), "attention_mask": tens(mask)}, "mean")[1]
    model = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=encoder_only, dtype=dtype)).to(DEV).eval()
    _, got = model.encode_passage({"input_ids": tens(ids).to(DEV), "attention_mask": tens(mask).to(DEV)})
    got = got.float().cpu()
    if dtype == "float32":
        assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    elif dtype == "float16":
        assert torch.nn.functional.cosine_similarity(got, ref, dim=1).min().item() > 0.99999
    else:
        assert torch.nn.functional.cosine_similarity(got, ref, dim=1).min().item() > 0.999


def test_float32_training_rejects_more_than_192_tokens():
    """The float32 backward attention kernel holds three [64][L + 4] f32 images in LDS: 192 keys is what 160 KiB takes.
    Longer float32 batches must fail with a message, not a HIP error (and leave no sticky error behind)."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch_amd.native import NativeError
    torch.manual_seed(5)
    cfg = BertConfig(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                     max_position_embeddings=256, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    rng = np.random.default_rng(0)
    p_ids, p_mask = synth_tokens(rng, 4, 224, vocab=600, lo_len=200, lo_id=300)
    q_ids, q_mask = synth_tokens(rng, 2, 32, vocab=600, lo_len=4, lo_id=300)
    tens = lambda a: torch.from_numpy(a).to(DEV)
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=False, dtype="float32"),
                    data_args=NS(train_n_passages=2),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=2)).to(DEV).train()
    with pytest.raises((NativeError, RuntimeError), match="192"):
        out = model(query={"input_ids": tens(q_ids), "attention_mask": tens(q_mask)},
                    passage={"input_ids": tens(p_ids), "attention_mask": tens(p_mask)})
        out.loss.backward()
    torch.cuda.synchronize()
    assert torch.ones(4, device=DEV).sum().item() == 4.0          # the device and torch's error state are fine


def _pair_dataset(g, n=8):
    """Pre-collated (query batch, passage batch) items, as QPCollator would hand them to the trainer."""
    q = {"input_ids": torch.from_numpy(g["q_input_ids"]), "attention_mask": torch.from_numpy(g["q_attention_mask"])}
    p = {"input_ids": torch.from_numpy(g["p_input_ids"]), "attention_mask": torch.from_numpy(g["p_attention_mask"])}
    return [(q, p)] * n


def _trainer_args(tmp_path, **kw):
    base = dict(device=DEV, world_size=1, process_index=0, per_device_train_batch_size=1, dataloader_num_workers=0,
                dataloader_pin_memory=False, negatives_x_device=False, learning_rate=1e-3, weight_decay=0.0,
                adam_beta1=0.9, adam_beta2=0.999, adam_epsilon=1e-8, warmup_ratio=0.1, warmup_steps=0, max_steps=12,
                num_train_epochs=1, gradient_accumulation_steps=1, max_grad_norm=1.0, logging_steps=4, save_steps=0,
                output_dir=str(tmp_path), fp16=False, bf16=False, seed=1, gc_q_chunk_size=2, gc_p_chunk_size=4)
    base.update(kw)
    return NS(**base)


def test_drtrainer_loop_reduces_loss_and_saves(golden, tmp_path):
    from openmatch.trainer import DRTrainer
    g = golden("train_bert_tiny")
    model = _train_model(g)
    # with validation loss during training (reference driver/train_dr.py:84-97, docs/dr-msmarco-passage.md:71-85)
    args = _trainer_args(tmp_path, evaluation_strategy="steps", eval_steps=6, per_device_eval_batch_size=1)
    trainer = DRTrainer(model=model, args=args, train_dataset=_pair_dataset(g), eval_dataset=_pair_dataset(g, 3),
                        data_collator=lambda b: b[0])
    before = trainer.evaluate()
    assert abs(before["eval_loss"] - float(g["loss"])) < 1e-4 and before["eval_samples"] == 3 * g["q_input_ids"].shape[0]
    trainer.train()
    assert model.training                                                      # evaluate() restores the mode it found
    evals = [h for h in trainer.state.log_history if "eval_loss" in h]
    assert [h["step"] for h in evals] == [0, 6, 12] and evals[-1]["eval_loss"] < evals[1]["eval_loss"] < evals[0]["eval_loss"]
    hist = [h for h in trainer.state.log_history if "loss" in h]
    assert trainer.state.global_step == 12 and len(hist) == 3
    assert hist[-1]["loss"] < hist[0]["loss"] < float(g["loss"]) + 0.05       # same batch every step: it must fit
    trainer.save_model()
    assert {"openmatch_config.json", "linear.pt", "head_config.json", "training_args.bin"} <= set(__import__("os").listdir(tmp_path))


def test_overlapped_bucketed_allreduce_leaves_single_rank_gradients_unchanged(golden):
    """openmatch_amd/grad_sync.py on the GPU: per-layer events recorded by om_encoder_train_backward, buckets reduced on a
    side stream while the backward runs.  On a one-rank RCCL group the mean all-reduce is the identity, so the gradients
    must equal those of a plain backward -- which exercises the event hand-off, the bucket slicing of the real arena and
    the stream ordering (the arithmetic across ranks is covered over gloo in tests/test_distributed_cpu.py)."""
    import torch.distributed as dist
    from openmatch_amd.grad_sync import GradSync
    from openmatch_amd.trainer.dense_trainer import allreduce_mean_
    g = golden("train_bert_tiny")
    q, p = _train_batch(g)
    plain = _train_model(g)
    plain(query=q, passage=p).loss.backward()
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        model = _train_model(g)
        sync = GradSync(1, bucket_layers=1)
        out = model(query=q, passage=p)
        sync.begin()
        out.loss.backward()
        sync.finish()
        assert len(sync.reduced) >= 1 and not sync.works        # (two arenas when queries and passages take separate passes)
        params = [t for t in model.parameters() if t.grad is not None]
        allreduce_mean_(params, 1, skip_storages=sync.reduced)
        torch.cuda.synchronize()
        for (n, a), (_, b) in zip(model.named_parameters(), plain.named_parameters()):
            if b.grad is not None:
                # (the weight-gradient kernel adds its split-M partials with f32 atomics: two backward passes agree to rounding)
                assert torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-6), n
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_layers,wbatch", [(2, 4), (4, 4), (1, 2), (4, 0)])
def test_bucket_handover_in_bf16_waits_for_the_deferred_weight_gradients(bucket_layers, wbatch):
    """The path the trainer actually runs: bfloat16, widths of 256, M >= 32 -- the backward keeps the layers' activation
    gradients, launches the weight gradients of a GROUP of layers in one batched kernel on its side lane and records the layers'
    events behind it.  A bucket handed to the collective before that launch has written its slice would be reduced too early; on
    one device an all-reduce cannot show it, so the collective is replaced by a visible in-place operation (x 2 on the side
    stream): every gradient must come out as twice the plain backward's, for bucket sizes below, equal to and above the
    weight-gradient group (and with the group launch off)."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch_amd import native as N
    from openmatch_amd.grad_sync import GradSync
    torch.manual_seed(3)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=6, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    mk_model = lambda: DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="bfloat16"),
                               data_args=NS(train_n_passages=4),
                               train_args=NS(negatives_x_device=False, per_device_train_batch_size=4)).to(DEV).train()
    g = torch.Generator().manual_seed(4)
    mk = lambda n, L: {"input_ids": torch.randint(300, 600, (n, L), generator=g).to(DEV), "attention_mask": torch.ones(n, L, dtype=torch.long).to(DEV)}
    q, p = mk(4, 16), mk(16, 48)
    N.check(N.lib().om_debug_option(N.OPT_TRAIN_WGRAD_BATCH, wbatch))
    try:
        model = mk_model()
        model(query=q, passage=p).loss.backward()
        plain = {n: t.grad.clone() for n, t in model.named_parameters() if t.grad is not None}
        model.zero_grad(set_to_none=True)
        sync = GradSync(1, bucket_layers=bucket_layers, collective=lambda flat: flat.mul_(2.0))
        out = model(query=q, passage=p)
        sync.begin()
        try:
            out.loss.backward()
        finally:
            sync.finish()
        torch.cuda.synchronize()
        assert len(sync.reduced) == 1                       # the tied training step takes ONE pass: one arena
        n_checked = 0
        for n, t in model.named_parameters():
            if n in plain:
                # (a bucket handed over early leaves part of it at 1x; sums formed with atomics -- the embedding scatter -- differ
                #  between two backward passes in the last bits only)
                assert torch.allclose(t.grad, 2.0 * plain[n], rtol=1e-3, atol=1e-6), (n, (t.grad - 2.0 * plain[n]).abs().max().item())
                n_checked += 1
        assert n_checked > 6 * 12
    finally:
        N.check(N.lib().om_debug_option(N.OPT_TRAIN_WGRAD_BATCH, 4))


def test_cross_device_negatives_over_a_one_rank_rccl_group_equal_the_local_step(golden):
    """DRModel with negatives_x_device on GPU tensors through RCCL (dist_gather_tensor: one all_gather_into_tensor, the local
    slot keeping autograd; reference modeling/dense_retrieval_model.py:247-258, loss.py:33-38).  With one rank the gathered
    batch IS the local one: loss, scores and every gradient must equal the plain step's.  (Two ranks: gloo, tests/test_distributed_cpu.py.)"""
    import torch.distributed as dist
    from openmatch.modeling import DRModel, LinearHead
    g = golden("train_bert_tiny")
    q, p = _train_batch(g)
    plain = _train_model(g)
    out0 = plain(query=q, passage=p)
    out0.loss.backward()
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29549", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        cfg, lm = model_from_golden(g, "bert", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        head = LinearHead(128, 128)
        head.linear.weight.data.copy_(torch.from_numpy(g["head_w"]))
        model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                        model_args=NS(encoder_only=False, dtype="float32"), data_args=NS(train_n_passages=int(g["n_psg"])),
                        train_args=NS(negatives_x_device=True, per_device_train_batch_size=4)).to(DEV).train()
        out1 = model(query=q, passage=p)
        out1.loss.backward()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert abs(out1.loss.item() - out0.loss.item()) < 1e-6
    assert torch.allclose(out1.scores, out0.scores, atol=1e-6)
    for (n, a), (_, b) in zip(model.named_parameters(), plain.named_parameters()):
        if b.grad is not None:
            assert a.grad is not None and torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-6), n


def _ragged_train_batch(rng, B, L, vocab=600):
    ids, mask = synth_tokens(rng, B, L, vocab=vocab, lo_len=3, lo_id=300)
    ids[0, :], mask[0, :] = rng.integers(300, vocab, L), 1        # a full-length row
    mask[1, :] = 0; mask[1, ::3] = 1                              # holes
    mask[2, :] = 0                                                # a row without any token
    return torch.from_numpy(ids), torch.from_numpy(mask)


@pytest.mark.parametrize("L", [128, 200])
@pytest.mark.parametrize("pooling", ["first", "mean"])
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_packed_rows_training_step_matches_the_padded_step(dtype, pooling, L):
    """Round 5: om_encoder_train_forward_packed / _backward_packed run the training step over the tokens up to each sequence's last
    unmasked one instead of B x L (the reference pads and computes over the padding, dataset/data_collator.py:13-24).  Same
    representations and the same gradient for EVERY parameter as the padded pair, up to the order of 16-bit-sized sums: ragged
    lengths, a full row, a mask with holes, an empty row; a row bound below the token count poisons the step (NaN) instead of
    truncating it; with dropout (same seed) the packed step equals the padded step as well: the masks are keyed on the token."""
    from transformers import BertConfig, BertModel
    from openmatch_amd import train as T
    from openmatch_amd.encoder import compute_dtype_code, rows_bound_of, token_rows_of
    torch.manual_seed(41)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=256, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg).to(DEV).train()
    rng = np.random.default_rng(6)
    B = 24                       # (L = 200: the attention backward's generic kernel; 128: the transposing-read one)
    ids, mask = _ragged_train_batch(rng, B, L)
    tokens = int(token_rows_of(mask).sum())
    rows = rows_bound_of(token_rows_of(mask))
    assert rows is not None and rows == rows_bound_of(tokens) and rows < B * L
    items = {"input_ids": ids.to(DEV), "attention_mask": mask.to(DEV)}
    code = compute_dtype_code(NS(dtype=dtype))
    wgt = torch.randn(B, 256, generator=torch.Generator().manual_seed(2)).to(DEV)
    wgt[2] = 0                       # (the empty row's representation is whatever a softmax over no keys gives: out of the loss)

    def step(packed_rows):
        lm.zero_grad(set_to_none=True)
        reps = T.encode_train(lm, None, items, pooling, False, code, True, packed_rows=packed_rows)[1]
        (reps * wgt).sum().backward()
        return reps.detach().clone(), {n: p.grad.detach().clone() for n, p in lm.named_parameters() if p.grad is not None}

    reps0, g0 = step(None)
    assert T.LAST_CALL == {"rows": B * L, "packed": False}
    reps1, g1 = step(rows)
    assert T.LAST_CALL == {"rows": rows, "packed": True}
    keep = torch.ones(B, dtype=torch.bool); keep[2] = False
    err = (reps1[keep] - reps0[keep]).abs().max().item() / reps0[keep].abs().max().item()
    assert torch.isfinite(reps1).all() and err < 2e-3, err
    assert set(g0) == set(g1)
    worst = ("", 0.0)
    for n in g0:
        a, b = g0[n].float(), g1[n].float()
        assert torch.isfinite(b).all(), n
        rel = ((a - b).norm() / a.norm().clamp_min(1e-12)).item()
        if "key.bias" in n:          # a softmax does not see a shift of its scores: the true gradient is ZERO, what is there is rounding noise
            assert b.norm().item() <= 1e-2 * g1[n.replace("key.bias", "query.bias")].norm().item() + 1e-6, n      # (bf16: 0.3 %)
            continue
        if a.norm().item() > 1e-6 and rel > worst[1]:
            worst = (n, rel)
    print(f"\n[packed training step, {dtype}, {pooling}] {tokens} tokens -> {rows} of {B * L} rows; reps max rel err {err:.2e}; worst gradient rel-L2 vs padded {worst[1]:.2e} ({worst[0]})")
    assert worst[1] < 2e-2, worst
    # a bound below the token count: NaN representations, never a truncated batch
    small = (tokens // 256) * 256 - 256
    if small >= 512:
        reps_bad, _ = step(small)
        assert T.LAST_CALL["packed"] and torch.isnan(reps_bad).all()
    # dropout (p = 0.1, both sites): every mask is keyed on (token, column) / (sequence, head, query, key), never on the packed row
    # (round 6; round 5 keyed the hidden-dropout masks on the row index, so the two entries drew different masks and the bench's
    # `train.ragged` printed two different losses for one batch): the packed step equals the padded step of the same seed exactly as
    # it does without dropout -- and differs from the deterministic step
    lm.config.hidden_dropout_prob = lm.config.attention_probs_dropout_prob = 0.1
    try:
        torch.manual_seed(1234)
        reps_d0, gd0 = step(None)
        assert T.LAST_CALL == {"rows": B * L, "packed": False}
        torch.manual_seed(1234)
        reps_d1, gd1 = step(rows)
        assert T.LAST_CALL == {"rows": rows, "packed": True}
    finally:
        lm.config.hidden_dropout_prob = lm.config.attention_probs_dropout_prob = 0.0
    assert torch.isfinite(reps_d1[keep]).all() and all(torch.isfinite(v).all() for v in gd1.values())
    assert not torch.equal(reps_d1, reps1)
    err_d = (reps_d1[keep] - reps_d0[keep]).abs().max().item() / reps_d0[keep].abs().max().item()
    worst_d = ("", 0.0)
    for n in gd0:
        if "key.bias" in n:
            continue
        a, b = gd0[n].float(), gd1[n].float()
        rel = ((a - b).norm() / a.norm().clamp_min(1e-12)).item()
        if a.norm().item() > 1e-6 and rel > worst_d[1]:
            worst_d = (n, rel)
    print(f"[packed training step, {dtype}, {pooling}, dropout 0.1] reps max rel err vs padded {err_d:.2e}; worst gradient rel-L2 vs padded {worst_d[1]:.2e} ({worst_d[0]})")
    assert err_d < 2e-3 and worst_d[1] < 2e-2, (err_d, worst_d)


@pytest.mark.parametrize("L", [96, 200])
@pytest.mark.parametrize("ff", ["relu", "gated-gelu"])
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_packed_rows_t5_training_step_matches_the_padded_step(dtype, ff, L, monkeypatch):
    """Round 6: the packed-rows training pair takes T5 encoder stacks too (VERDICT r5 "missing" 3).  Relative-position bias in the packed
    attention forward / backward (the table keeps the padded pitch), the bias gradient, RMSNorm, ReLU and gated-GELU feed-forwards, the
    shared embedding's scatter through the row map; a T5 block has no additive bias, so the rows no sequence owns stay exact zeros.
    Same representations and gradients as the padded pair without dropout AND with it (masks keyed on the token); mean and first pooling."""
    monkeypatch.delenv("OM_TRAIN_F16", raising=False); monkeypatch.delenv("OM_T5_F16", raising=False)
    from transformers import T5Config, T5EncoderModel
    from openmatch_amd import train as T
    from openmatch_amd.encoder import compute_dtype_code, rows_bound_of, token_rows_of
    torch.manual_seed(43)
    cfg = T5Config(d_model=256, d_ff=1024, num_layers=2, num_heads=4, d_kv=64, vocab_size=600, feed_forward_proj=ff, dropout_rate=0.0)
    lm = T5EncoderModel(cfg).to(DEV).train()
    rng = np.random.default_rng(8)
    B = 24
    ids, mask = _ragged_train_batch(rng, B, L)
    tokens = int(token_rows_of(mask).sum())
    rows = rows_bound_of(token_rows_of(mask))
    assert rows is not None and rows < B * L
    items = {"input_ids": ids.to(DEV), "attention_mask": mask.to(DEV)}
    code = compute_dtype_code(NS(dtype=dtype))
    wgt = torch.randn(B, 256, generator=torch.Generator().manual_seed(2)).to(DEV)
    wgt[2] = 0                       # (the empty row: out of the loss)
    keep = torch.ones(B, dtype=torch.bool); keep[2] = False
    for pooling in ("mean", "first"):
        def step(packed_rows):
            lm.zero_grad(set_to_none=True)
            reps = T.encode_train(lm, None, items, pooling, False, code, True, packed_rows=packed_rows)[1]
            (reps * wgt).sum().backward()
            return reps.detach().clone(), {n: p.grad.detach().clone() for n, p in lm.named_parameters() if p.grad is not None}

        def compare(drop):
            lm.config.dropout_rate = drop
            try:
                torch.manual_seed(77)
                r0, g0 = step(None)
                assert T.LAST_CALL == {"rows": B * L, "packed": False}
                torch.manual_seed(77)
                r1, g1 = step(rows)
                assert T.LAST_CALL == {"rows": rows, "packed": True}, T.LAST_CALL
            finally:
                lm.config.dropout_rate = 0.0
            err = (r1[keep] - r0[keep]).abs().max().item() / r0[keep].abs().max().item()
            assert torch.isfinite(r1[keep]).all() and set(g0) == set(g1)
            worst = ("", 0.0)
            for n in g0:
                a, b = g0[n].float(), g1[n].float()
                assert torch.isfinite(b).all(), n
                rel = ((a - b).norm() / a.norm().clamp_min(1e-12)).item()
                if a.norm().item() > 1e-6 and rel > worst[1]:
                    worst = (n, rel)
            return r1, err, worst

        r_det, err, worst = compare(0.0)
        r_drop, err_d, worst_d = compare(0.1)
        print(f"\n[packed T5 training step, {dtype}, {ff}, {pooling}, L={L}] {tokens} tokens -> {rows} of {B * L} rows; reps max rel err {err:.2e} "
              f"(dropout 0.1: {err_d:.2e}); worst gradient rel-L2 vs padded {worst[1]:.2e} ({worst[0]}), dropout 0.1: {worst_d[1]:.2e} ({worst_d[0]})")
        assert err < 2e-3 and worst[1] < 2e-2, (err, worst)
        assert err_d < 2e-3 and worst_d[1] < 2e-2, (err_d, worst_d)
        assert not torch.equal(r_det, r_drop)
    # a bound below the token count: NaN representations, never a truncated batch
    small = (tokens // 256) * 256 - 256
    if small >= 512:
        reps_bad = T.encode_train(lm, None, items, "mean", False, code, True, packed_rows=small)[1]
        assert T.LAST_CALL["packed"] and torch.isnan(reps_bad).all()


def test_trainer_takes_packed_rows_when_the_mask_is_still_on_the_host(golden, tmp_path, monkeypatch):
    """DRTrainer._prepare_inputs notes the batch's token count while the collator's mask is on the host; the one-pass training
    forward (queries padded to the passage length and encoded with the passages) then runs over the packed rows.  Same loss and
    gradients as with the switch off (OM_TRAIN_PACKED=0), to 16-bit noise; a batch that is already on the device keeps the padded pair."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch.trainer import DRTrainer
    from openmatch_amd import train as T
    torch.manual_seed(43)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    rng = np.random.default_rng(8)
    q_ids, q_mask = synth_tokens(rng, 4, 32, vocab=600, lo_len=4, lo_id=300)
    p_ids, p_mask = synth_tokens(rng, 32, 128, vocab=600, lo_len=10, lo_id=300)
    host = ({"input_ids": torch.from_numpy(q_ids), "attention_mask": torch.from_numpy(q_mask)},
            {"input_ids": torch.from_numpy(p_ids), "attention_mask": torch.from_numpy(p_mask)})
    res = {}
    for mode in ("packed", "off", "device"):
        model = DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="bfloat16"),
                        data_args=NS(train_n_passages=8), train_args=NS(negatives_x_device=False, per_device_train_batch_size=4)).to(DEV)
        model.zero_grad(set_to_none=True)
        monkeypatch.setenv("OM_TRAIN_PACKED", "0" if mode == "off" else "1")
        t = DRTrainer(model=model, args=_trainer_args(tmp_path), train_dataset=None)
        batch = host if mode != "device" else tuple({k: v.to(DEV) for k, v in b.items()} for b in host)
        loss = float(t.training_step(model, batch))
        assert T.LAST_CALL["packed"] == (mode == "packed"), (mode, T.LAST_CALL)
        res[mode] = (loss, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert res["packed"][0] == pytest.approx(res["off"][0], rel=2e-3, abs=2e-3) and res["device"][0] == pytest.approx(res["off"][0], rel=1e-6, abs=1e-6)
    for n, a in res["off"][1].items():
        b = res["packed"][1][n]
        assert ((a - b).norm() / a.norm().clamp_min(1e-12)).item() < 3e-2 or a.norm().item() < 1e-6, n
    # the gradient-cache trainer: the per-sequence token counts split with the chunks, so a chunk of 16 passages runs packed too;
    # its first, tape-less pass runs the TRAINING forward (a model in training mode always does), so the step differentiates the
    # representations the loss saw and lands on the full-batch step's gradients
    from openmatch.trainer import GCDenseTrainer
    gc = {}
    for mode in ("packed", "off"):
        monkeypatch.setenv("OM_TRAIN_PACKED", "0" if mode == "off" else "1")
        model = DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="bfloat16"),
                        data_args=NS(train_n_passages=8), train_args=NS(negatives_x_device=False, per_device_train_batch_size=4)).to(DEV)
        model.zero_grad(set_to_none=True)
        t = GCDenseTrainer(model=model, args=_trainer_args(tmp_path, gc_q_chunk_size=4, gc_p_chunk_size=16), train_dataset=None)
        loss = float(t.training_step(model, host))
        assert T.LAST_CALL["packed"] == (mode == "packed") and (mode == "off" or T.LAST_CALL["rows"] < 16 * 128)
        gc[mode] = (loss, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert gc["packed"][0] == pytest.approx(gc["off"][0], rel=1e-6, abs=1e-6) and gc["off"][0] == pytest.approx(res["off"][0], rel=2e-3, abs=2e-3)
    for n, a in gc["off"][1].items():
        b = gc["packed"][1][n]
        assert ((a - b).norm() / a.norm().clamp_min(1e-12)).item() < 1e-3 or a.norm().item() < 1e-6, n
    # against the full-batch step: the chunks' contractions run at other row counts (other tile kernels) than the one-pass batch's, so
    # the representations agree to bfloat16's last digits only -- which this random-init model's near-equal scores amplify tensor by
    # tensor (8 % on the word embeddings, more on sums that cancel); the step as a whole points the same way
    names = sorted(res["off"][1])
    flat = lambda d: torch.cat([d[n].float().reshape(-1) for n in names])
    cosine = torch.nn.functional.cosine_similarity(flat(gc["off"][1]), flat(res["off"][1]), dim=0).item()
    print(f"\n[gradient cache vs full batch, bf16, random-init] cosine of the whole gradient {cosine:.4f}")
    assert cosine > 0.97, cosine


def test_step_eval_cycles_keep_one_fold_buffer(tmp_path):
    """FusedAdamW refreshes the packed 16-bit weights in place and drops what was derived from them (the
    LayerNorm-folded inference weights); an evaluation between two steps folds again.  The fold must go back into the SAME device
    buffer: step / eval cycles keep the packed object's buffer list and the allocated device memory constant, and every
    evaluation sees the updated weights."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModel
    from openmatch.trainer import DRTrainer
    from openmatch_amd import encoder as enc
    torch.manual_seed(47)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                     max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    rng = np.random.default_rng(9)
    q_ids, q_mask = synth_tokens(rng, 4, 32, vocab=600, lo_len=4, lo_id=300)
    p_ids, p_mask = synth_tokens(rng, 32, 128, vocab=600, lo_len=10, lo_id=300)
    dev = lambda i, m: {"input_ids": torch.from_numpy(i).to(DEV), "attention_mask": torch.from_numpy(m).to(DEV)}
    q, p = dev(q_ids, q_mask), dev(p_ids, p_mask)
    model = DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False, dtype="bfloat16"),
                    data_args=NS(train_n_passages=8), train_args=NS(negatives_x_device=False, per_device_train_batch_size=4)).to(DEV)
    t = DRTrainer(model=model, args=_trainer_args(tmp_path, learning_rate=1e-2), train_dataset=None)
    t.create_optimizer_and_scheduler(num_training_steps=100)
    assert type(t.optimizer).__name__ == "FusedAdamW"

    def evaluate():
        model.eval()
        with torch.no_grad():
            reps = model(passage=p).p_reps.float().clone()       # 32 x 128 = 4096 tokens: the fused (LayerNorm-folded) path
        model.train()
        return reps

    def inference_pk():
        cache = lm.__dict__[enc._PACK_CACHE_ATTR]
        pks = [v[1] for v in cache.values() if getattr(v[1], "fold_blob", None) is not None]
        assert len(pks) == 1, len(pks)
        return pks[0]

    state, prev = [], None
    for cycle in range(5):
        model.zero_grad(set_to_none=True)
        t.training_step(model, (q, p))
        t.optimizer_step()
        reps = evaluate()
        assert torch.isfinite(reps).all()
        if prev is not None:
            assert not torch.equal(reps, prev), "the evaluation must see the step's weights"
        prev = reps
        pk = inference_pk()
        torch.cuda.synchronize()
        state.append((len(pk.keep), pk.fold_blob.data_ptr(), torch.cuda.memory_allocated(DEV)))
    print(f"\n[step/eval cycles] (len(keep), fold buffer, bytes allocated) per cycle: {state}")
    assert len({s[0] for s in state[1:]}) == 1 and len({s[1] for s in state[1:]}) == 1, state
    assert state[-1][2] <= state[1][2] + (1 << 20), state


@pytest.mark.parametrize("fp16", [False, True])
def test_gradient_cache_step_equals_full_batch_step(golden, tmp_path, fp16):
    """GCDenseTrainer (chunked, re-encoded) must give the SAME gradients as one full-batch step; under --fp16 (float16 kernels, the
    loss scale carried into every chunk's backward through the cached representation gradients) the same to 16-bit noise."""
    from openmatch.trainer import DRTrainer, GCDenseTrainer
    g = golden("train_bert_tiny")
    batch = _pair_dataset(g, 1)[0]
    grads = []
    for cls in (DRTrainer, GCDenseTrainer):
        model = _train_model(g)
        t = cls(model=model, args=_trainer_args(tmp_path, fp16=fp16, fp16_init_scale=1024.0), train_dataset=None)
        loss = t.training_step(model, batch)
        assert abs(float(loss) - float(g["loss"])) < (2e-3 if fp16 else 1e-5)
        inv = 1.0 / float(t._loss_scaler().state[0]) if fp16 else 1.0
        assert not fp16 or inv == 1.0 / 1024.0
        grads.append({n: p.grad.clone() * inv for n, p in model.named_parameters() if p.grad is not None})
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        if fp16:
            assert (a - b).norm() <= 2e-2 * a.norm() + 1e-7, (n, ((a - b).norm() / a.norm()).item())
        else:
            assert (a - b).abs().max() <= 1e-6 + 1e-4 * a.abs().max(), n


# ------------------------------------------------------------------------------- cross-encoder (config 5)
def test_cross_encoder_bert_large_width_matches_oracle():
    """RRModel scoring at BASELINE config 5's shape: H = 1024, 16 heads, F = 4096, L = 162 pairs with
    token types 0/1 (2 layers, random weights): scores vs the CPU oracle, then Reranker.rerank order."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import LinearHead, RRModel
    from openmatch.retriever import Reranker
    torch.manual_seed(21)
    cfg = BertConfig(hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                     vocab_size=600, max_position_embeddings=192)
    lm = BertModel(cfg).eval()
    head = LinearHead(1024, 1)
    sd = {k: v.clone() for k, v in lm.state_dict().items()}
    hw = head.linear.weight.detach().clone()
    rng = np.random.default_rng(9)
    n = 24
    ids, mask = synth_tokens(rng, n, 162, vocab=600, lo_len=20, lo_id=300)
    tt = np.zeros_like(ids)
    for i in range(n):
        ln = int(mask[i].sum()); tt[i, ln // 3:ln] = 1
    items = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.from_numpy(tt)}
    _, ref = encoder_ref.encode(sd, cfg, "bert", items, "first", hw, False)
    model = RRModel(lm=lm, head=head, pooling="first", model_args=NS(encoder_only=False, dtype="float32"))
    args = NS(device=DEV, world_size=1, process_index=0, local_process_index=0, fp16=False, eval_batch_size=8,
              per_device_eval_batch_size=8, dataloader_num_workers=0, dataloader_pin_memory=False)
    rr = Reranker(model, None, None, args)
    with torch.no_grad():
        got = rr.model.encode({k: v.to(DEV) for k, v in items.items()})
    assert got.shape == (n, 1)
    assert np.abs(got.cpu().numpy() - ref.numpy()).max() < 1e-4 * max(1.0, np.abs(ref.numpy()).max())
    pairs = [{"query_id": f"q{i // 8}", "doc_id": f"d{i}", "input_ids": ids[i], "attention_mask": mask[i],
              "token_type_ids": tt[i]} for i in range(n)]
    run = rr.rerank(None, None, pair_dataset=pairs)
    for q in range(3):
        want = sorted(range(q * 8, q * 8 + 8), key=lambda i: -float(ref[i, 0]))
        have = [int(d[1:]) for d, _ in sorted(run[f"q{q}"].items(), key=lambda kv: -kv[1])]
        assert have == want


@pytest.mark.parametrize("max_norm", [0.0, 0.5])
def test_fused_adamw_matches_torch_adamw(max_norm):
    """openmatch_amd.optim.FusedAdamW (om_grad_sqnorm + om_adamw_step) against clip_grad_norm_ + torch.optim.AdamW (the
    single-tensor reference form) over five steps: odd sizes (chunk tails, unaligned tails), two weight-decay groups, a
    parameter that stops receiving gradients, a changing learning rate."""
    from openmatch_amd.optim import FusedAdamW
    gen = torch.Generator().manual_seed(3)
    sizes = [(1,), (7,), (16384,), (16385,), (257, 389), (3, 5, 7), (40000,)]
    mine = [torch.nn.Parameter(torch.randn(*s, generator=gen).to(DEV)) for s in sizes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    mk = lambda ps: [{"params": ps[:4], "weight_decay": 0.01}, {"params": ps[4:], "weight_decay": 0.0}]
    fo = FusedAdamW(mk(mine), lr=1e-2, betas=(0.9, 0.98), eps=1e-8, max_grad_norm=max_norm)
    to = torch.optim.AdamW(mk(ref), lr=1e-2, betas=(0.9, 0.98), eps=1e-8, foreach=False, fused=False)
    for step in range(5):
        for i, (a, b) in enumerate(zip(mine, ref)):
            if i == 2 and step >= 3:                     # no gradient any more: both leave it alone
                a.grad = b.grad = None
                continue
            gr = (torch.randn(*a.shape, generator=gen) * (3.0 if step == 1 else 0.3)).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        for o in (fo, to):
            for grp in o.param_groups:
                grp["lr"] = 1e-2 / (1 + step)
        if max_norm > 0:
            tn = torch.nn.utils.clip_grad_norm_([p for p in ref if p.grad is not None], max_norm)
        fo.step(); to.step()
        if max_norm > 0:
            assert abs(fo.grad_norm().item() - tn.item()) <= 1e-5 * tn.item()
        for a, b in zip(mine, ref):
            assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item()), (step, tuple(a.shape))
            sa, sb = fo.state[a], to.state[b]
            if "exp_avg" in sb:
                assert (sa["exp_avg"] - sb["exp_avg"]).abs().max().item() <= 2e-6 * max(1.0, sb["exp_avg"].abs().max().item())
                assert (sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max().item() <= 2e-6 * max(1.0, sb["exp_avg_sq"].abs().max().item())
                assert sa["step"] == int(sb["step"])
    # a non-finite gradient with skip_nonfinite: nothing moves (the skipped step of a float16 loss scaler)
    fo.skip_nonfinite = True
    before = [p.detach().clone() for p in mine]
    for p in mine:
        p.grad = torch.zeros_like(p)
    mine[4].grad[3, 5] = float("inf")
    fo.step()
    assert not math.isfinite(fo.grad_norm().item())
    assert all(torch.equal(a, b) for a, b in zip(mine, before))


@pytest.mark.parametrize("arch", ["bert", "t5"])
@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_few_rows_forward_tracks_f32_path_and_is_batch_invariant(dtype, arch):
    """Forwards of up to OM_OPT_GEMM_SKINNY_M (1 024) token rows -- a served query, a handful of sequences -- run their contractions on
    the weight-streaming kernel with the normalisations as kernels (encoder.hip few_rows).  Against the exact-f32 HIP path at 1 ... 1 024
    rows; the same numbers (to 16-bit noise) as the tile kernels give with the path switched off; and a sequence encoded ALONE gives
    the same bits as the same sequence inside a batch (the K split of the kernel depends on the weight's shape only)."""
    from transformers import BertConfig, BertModel, T5Config, T5EncoderModel
    from openmatch.modeling import DRModelForInference
    from openmatch_amd import native as N_
    torch.manual_seed(31)
    if arch == "bert":
        lm = BertModel(BertConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=1024, vocab_size=600,
                                  max_position_embeddings=256)).eval()
    else:
        lm = T5EncoderModel(T5Config(d_model=256, d_ff=1024, num_layers=3, num_heads=4, d_kv=64, vocab_size=600, feed_forward_proj="relu")).eval()
    mk = lambda dt: DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", model_args=NS(encoder_only=arch == "t5", dtype=dt)).to(DEV).eval()
    m16, m32 = mk(dtype), mk("float32")
    rng = np.random.default_rng(12)
    tol = 5e-6 if dtype == "float16" else 2e-4
    worse = []
    for B, L in ((1, 32), (1, 7), (3, 32), (8, 32), (5, 128), (8, 128), (31, 33)):
        ids, mask = synth_tokens(rng, B, L, vocab=600, lo_len=max(2, L // 3), lo_id=300)
        ids[0, :], mask[0, :] = rng.integers(300, 600, L), 1
        items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
        few = m16.encode_passage(items)[1].double().cpu()
        ref = m32.encode_passage(items)[1].double().cpu()
        N_.check(N_.lib().om_debug_option(19, 0))
        try:
            tiles = m16.encode_passage(items)[1].double().cpu()
        finally:
            N_.check(N_.lib().om_debug_option(19, 1024))
        cos = lambda a, b: torch.nn.functional.cosine_similarity(a, b, dim=1).min().item()
        assert 1 - cos(few, ref) < tol and 1 - cos(tiles, ref) < tol, (B, L, cos(few, ref), cos(tiles, ref))
        assert 1 - cos(few, ref) <= 2.0 * (1 - cos(tiles, ref)) + 1e-7, (B, L, cos(few, ref), cos(tiles, ref))
        # BERT (round 6): the few-rows path keeps its residual stream in f32 as the reference's autocast does (the tile kernels below
        # 512 rows keep ONE 16-bit plane): measured by the relative row error, it is the closer of the two at every batch of the list
        rel = lambda a: ((a - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
        worse.append((B, L, rel(few), rel(tiles)))
        if dtype == "bfloat16" and arch == "bert" and B * L >= 512:
            continue            # bfloat16 BERT from 512 rows on keeps the fused path (two-plane residual stream): another code path than L rows alone
        alone = m16.encode_passage({k: v[:1] for k, v in items.items()})[1]
        batch = m16.encode_passage(items)[1]
        assert torch.equal(alone[0], batch[0]), (B, L)
        if arch == "bert" and B * L <= 64:      # pending LayerNorms (round 6, <= OM_OPT_FEW_ROWS_LN_FUSE rows) == the LayerNorm kernels, bit for bit
            N_.check(N_.lib().om_debug_option(21, 0))
            try:
                with_kernels = m16.encode_passage(items)[1]
            finally:
                N_.check(N_.lib().om_debug_option(21, 64))
            assert torch.equal(with_kernels, batch), (B, L)
    print(f"\n[few rows, {arch}, {dtype}] max relative row error vs f32 (few-rows path / tile kernels): " + ", ".join(f"{b}x{l}: {a:.1e} / {t:.1e}" for b, l, a, t in worse))
    if arch == "bert":
        few_sum, tile_sum = sum(a for _, _, a, _ in worse), sum(t for _, _, _, t in worse)
        assert few_sum < tile_sum, (few_sum, tile_sum)


def test_index_scan_on_16x16x32_mfmas_returns_the_lists_of_the_default_kernel():
    """OM_GEMM_CONT bit 10 (round 6, opt-in: a wash end to end, profiles/r06_scan_16x16x32_ab.txt): the wide-batch scan on the encoder's
    16 x 16 x 32 K loop.  Same scores, same ids as the default 32 x 32 x 16 kernel, k-way ties included, on a ragged row count."""
    from openmatch_amd import native as N_
    from openmatch_amd.index import FlatIPIndex
    g = torch.Generator().manual_seed(9)
    N, nq, d, k = 70000 + 131, 300, 768, 200
    rows = torch.randn(N, d, generator=g)
    rows[1000:1040] = rows[999]                      # a 41-way exact tie
    q = torch.randn(nq, d, generator=g)
    q[7] = rows[999] * 3.0                           # ... that one query ranks first
    idx = FlatIPIndex(d, device=DEV)
    idx.add(rows.numpy())
    D0, I0 = idx.search(q.numpy(), k)
    N_.check(N_.lib().om_debug_option(16, 495 | 1024))
    try:
        D1, I1 = idx.search(q.numpy(), k)
    finally:
        N_.check(N_.lib().om_debug_option(16, 495))
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)


def test_layernorm_row_reduction_without_the_lds_crossbar_equals_the_shuffle_butterfly():
    """ln_row.h sums a row over the 64 lanes with v_permlane32/16_swap, DPP row rotation, ds_swizzle and DPP quad permutations in the
    pairing of the __shfl_xor butterfly (common.h wave_sum): the same bits, on values of mixed sign and magnitude, in every lane."""
    from openmatch_amd import native as N_
    g = torch.Generator().manual_seed(3)
    groups = 4099
    x = (torch.randn(groups * 64, generator=g) * torch.exp(torch.randn(groups * 64, generator=g) * 3)).to(DEV)
    a, b = torch.zeros(groups, device=DEV), torch.zeros(groups, device=DEV)
    N_.check(N_.lib().om_debug_wave_sum_check(x.data_ptr(), a.data_ptr(), b.data_ptr(), groups, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert torch.allclose(a.double(), x.view(groups, 64).double().sum(1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_pending_layernorms_of_a_handful_of_rows_give_the_bits_of_the_layernorm_kernels(dtype):
    """Round 6: a 16-bit BERT forward over <= OM_OPT_FEW_ROWS_LN_FUSE (64) token rows launches no LayerNorm kernel between the embedding and
    the last layer -- the contraction that consumes LN(y) normalises its operand rows itself, the one that adds LN(y) re-derives the element
    from (mean, rstd) (gemm_skinny.hip; ln_row.h holds the one definition of the arithmetic).  At bert-base width (three 4-element vectors
    per lane and row), pooled representations AND the hidden states equal the path with the normalisations as kernels bit for bit; mean
    and first pooling; 1 ... 64 rows."""
    from transformers import BertConfig, BertModel
    from openmatch.modeling import DRModelForInference
    from openmatch_amd import native as N_
    torch.manual_seed(33)
    lm = BertModel(BertConfig(num_hidden_layers=3, vocab_size=2000, max_position_embeddings=128)).eval()
    rng = np.random.default_rng(5)
    for pooling in ("first", "mean"):
        m16 = DRModelForInference(lm_q=lm, lm_p=lm, pooling=pooling, model_args=NS(encoder_only=False, dtype=dtype)).to(DEV).eval()
        m32 = DRModelForInference(lm_q=lm, lm_p=lm, pooling=pooling, model_args=NS(encoder_only=False, dtype="float32")).to(DEV).eval()
        for B, L in ((1, 32), (1, 5), (2, 32), (3, 17), (1, 64), (4, 16)):
            ids, mask = synth_tokens(rng, B, L, vocab=2000, lo_len=max(2, L // 2), lo_id=1000)
            items = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
            hid, reps = m16.encode_passage(items)
            N_.check(N_.lib().om_debug_option(21, 0))
            try:
                hid_k, reps_k = m16.encode_passage(items)
            finally:
                N_.check(N_.lib().om_debug_option(21, 64))
            assert torch.equal(reps, reps_k), (pooling, B, L, (reps - reps_k).abs().max().item())
            assert torch.equal(hid, hid_k), (pooling, B, L)
            ref = m32.encode_passage(items)[1]
            assert 1 - torch.nn.functional.cosine_similarity(reps.double(), ref.double(), dim=1).min().item() < (5e-6 if dtype == "float16" else 2e-4)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_few_row_contractions_on_the_weight_streaming_kernel(dtype):
    """gemm_skinny.hip (om_gemm_nt with M <= OM_OPT_GEMM_SKINNY_M rows, 16-bit): every epilogue the small forwards use -- bias,
    erf-GELU / ReLU / tanh-GELU, residual added or multiplied (in place too) -- against torch in f32 on the same 16-bit operands;
    a row's bits do not depend on how many rows ride along; switched off, the tile kernels give the same numbers."""
    from openmatch_amd import native as N_
    td = getattr(torch, dtype)
    code = N_.OM_BF16 if dtype == "bfloat16" else N_.OM_F16
    lib = N_.lib()
    gen = torch.Generator().manual_seed(17)
    tol = 2e-2 if dtype == "bfloat16" else 3e-3

    def run(A, W, bias, resid, act, M=None, inplace=False):
        M = A.shape[0] if M is None else M
        C = resid.clone() if inplace else torch.empty(A.shape[0], W.shape[0], device=DEV, dtype=td)
        r = C if inplace else resid
        with torch.cuda.device(DEV):
            N_.check(lib.om_gemm_nt(code, N_.ptr(A), A.shape[1], N_.ptr(W), W.shape[1], code, N_.ptr(C), W.shape[0], M, W.shape[0], A.shape[1],
                                    N_.ptr(bias) if bias is not None else None, N_.ptr(r) if r is not None else None, W.shape[0], act,
                                    N_.stream_ptr(torch.device(DEV))))
        return C

    acts = {N_.ACT_NONE: lambda v: v, N_.ACT_GELU_ERF: torch.nn.functional.gelu, N_.ACT_RELU: torch.relu,
            N_.ACT_GELU_TANH: lambda v: torch.nn.functional.gelu(v, approximate="tanh")}
    for (M, Nn, K) in [(1, 768, 768), (5, 2304, 768), (32, 3072, 768), (33, 768, 3072), (64, 256, 256), (100, 768, 1024), (256, 3072, 768)]:
        A = (torch.randn(M, K, generator=gen) * 0.5).to(DEV, td)
        W = (torch.randn(Nn, K, generator=gen) * 0.05).to(DEV, td)
        bias = torch.randn(Nn, generator=gen).to(DEV)
        resid = torch.randn(M, Nn, generator=gen).to(DEV, td)
        base = A.float() @ W.float().t()
        for act, fn in acts.items():
            for mode in ("plain", "add", "mul", "inplace"):
                if mode == "mul" and act != N_.ACT_GELU_TANH:
                    continue
                want = fn(base + bias)
                if mode in ("add", "inplace"):
                    want = want + resid.float()
                elif mode == "mul":
                    want = want * resid.float()
                got = run(A, W, bias, None if mode == "plain" else resid, act | (N_.ACT_MUL_RESID if mode == "mul" else 0), inplace=mode == "inplace")
                err = (got.float() - want).abs().max().item() / max(1.0, want.abs().max().item())
                assert err < tol, (M, Nn, K, act, mode, err)
        full = run(A, W, bias, resid, N_.ACT_GELU_ERF)
        one = run(A, W, bias, resid, N_.ACT_GELU_ERF, M=1)
        assert torch.equal(one[:1], full[:1])                      # row 0 alone == row 0 of the batch, bit for bit
        N_.check(lib.om_debug_option(19, 0))                        # OM_OPT_GEMM_SKINNY_M = 0: the tile kernels
        try:
            tiles = run(A, W, bias, resid, N_.ACT_GELU_ERF)
        finally:
            N_.check(lib.om_debug_option(19, 1024))
        assert (tiles.float() - full.float()).abs().max().item() <= 2 * tol * max(1.0, full.float().abs().max().item())


def test_fused_adamw_does_not_count_the_steps_the_loss_scaler_skipped():
    """torch.cuda.amp.GradScaler does not call optimizer.step() when the scaled gradients overflow, so AdamW's bias corrections
    count the steps TAKEN.  FusedAdamW + LossScaler keep the skip on the device (no host read): two overflowing steps, then
    two clean ones, must leave the parameters and both moments where torch.optim.AdamW's first two steps on the same (unscaled)
    gradients leave them, and the scale halved twice."""
    from openmatch_amd.optim import FusedAdamW, LossScaler
    gen = torch.Generator().manual_seed(5)
    sizes = [(33,), (16385,), (64, 96)]
    mine = [torch.nn.Parameter(torch.randn(*s, generator=gen).to(DEV)) for s in sizes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    fo = FusedAdamW(mine, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0)
    to = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, foreach=False, fused=False)
    sc = LossScaler(DEV, init_scale=1024.0, growth_interval=1000)
    sc.attach(fo)
    start = [p.detach().clone() for p in mine]
    for _ in range(2):                                             # overflow: skipped, scale halves
        for p in mine:
            p.grad = torch.full_like(p, float("inf"))
        fo.step(); sc.update(fo)
    assert all(torch.equal(a, b) for a, b in zip(mine, start))
    assert sc.skipped_steps() == 2 and float(sc.state[0]) == 256.0
    for _ in range(2):
        for a, b in zip(mine, ref):
            gr = torch.randn(*a.shape, generator=gen).to(DEV)
            a.grad, b.grad = gr * sc.scale, gr.clone()             # what backward of (loss * scale) leaves
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        fo.step(); sc.update(fo); to.step()
        for a, b in zip(mine, ref):
            assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())
            for k in ("exp_avg", "exp_avg_sq"):
                assert (fo.state[a][k] - to.state[b][k]).abs().max().item() <= 2e-6 * max(1.0, to.state[b][k].abs().max().item()), k
    assert sc.skipped_steps() == 2 and float(sc.state[2]) == 2.0


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_optimizer_step_refreshes_the_packed_weights(golden, dtype):
    """The forward reads PACKED copies of the weights (Q|K|V fused, matrices in the compute dtype).  After DRTrainer.optimizer_step
    -- on FusedAdamW, which rewrites those copies in its own pass, and on torch's fused AdamW, which updates the parameters
    without bumping their versions (the trainer re-packs behind it) -- the next forward must see the NEW weights: its loss
    equals that of a freshly built model holding the same parameters."""
    import copy
    from openmatch.trainer import DRTrainer
    from openmatch_amd import encoder as enc
    g = golden("train_bert_tiny")
    q, p = _train_batch(g)
    for which in ("fused", "torch"):
        model = _train_model(g, dtype=dtype)
        args = NS(device=DEV, world_size=1, process_index=0, per_device_train_batch_size=4, negatives_x_device=False,
                  learning_rate=5e-2, weight_decay=0.01, max_grad_norm=1.0, gradient_accumulation_steps=1, fp16=False, bf16=False)
        trainer = DRTrainer(model=model, args=args)
        if which == "torch":
            trainer.optimizer = torch.optim.AdamW(model.parameters(), lr=5e-2, fused=True)
        else:
            trainer.create_optimizer_and_scheduler(num_training_steps=100)
            assert type(trainer.optimizer).__name__ == "FusedAdamW"
        l0 = trainer.training_step(model, (q, p)).item()
        trainer.optimizer_step()
        l1 = trainer.training_step(model, (q, p)).item()
        assert l1 < l0 - 1e-3, (which, l0, l1)                    # lr 5e-2 on one batch: the step is visible in the loss
        fresh = copy.deepcopy(model)                              # deep copies start with an empty pack cache
        with torch.no_grad():
            l_fresh = fresh(query=q, passage=p).loss.item()
        model.zero_grad(set_to_none=True)
        with torch.no_grad():
            l_again = model(query=q, passage=p).loss.item()
        assert l_again == l_fresh, (which, dtype, l_again, l_fresh)
        if which == "fused":                                      # the packed Q|K|V block is cat(q, k, v) of the UPDATED parameters
            at = model.lm_q.encoder.layer[0].attention.self
            sh = enc.shadows_of(at.key.weight)
            assert sh, "the fused Q|K|V buffer is registered as a copy of key.weight"
            pk, buf, off = sh[0]
            H = at.key.weight.shape[0]
            assert off == H * at.key.weight.shape[1]
            want = torch.cat([at.query.weight, at.key.weight, at.value.weight], 0).detach().to(buf.dtype)
            assert torch.equal(buf.view(3 * H, -1), want)


def test_float16_training_step_tracks_f32_and_scaler_skips_overflow(golden, tmp_path, monkeypatch):
    """float16 training (round 5): (1) the float16 kernels' gradients on the tiny fixture against the reference's fp32 gradients --
    direction and size per tensor, closer than the bfloat16 path's; dropout runs; (2) DRTrainer --fp16: the dynamic loss scale on
    the device -- an absurd initial scale overflows float16, those steps are skipped (parameters untouched) and the scale halves
    until the step goes through, then the loss falls as in the other formats."""
    monkeypatch.delenv("OM_TRAIN_F16", raising=False)          # the A/B switch that sends float16 training to the bfloat16 kernels
    from openmatch.trainer import DRTrainer
    from openmatch_amd import native as N
    g = golden("train_bert_tiny")
    q, p = _train_batch(g)
    errs = {}
    for dtype in ("float16", "bfloat16"):
        model = _train_model(g, dtype=dtype)
        out = model(query=q, passage=p)
        assert abs(out.loss.item() - float(g["loss"])) < (5e-3 if dtype == "float16" else 2e-2)
        (out.loss * 1024.0).backward()
        names = dict(model.lm_q.named_parameters())
        rels = []
        for key in g.files:
            if not key.startswith("g::") or key[3:] == "head_w":
                continue
            ref = torch.from_numpy(g[key]).flatten().double()
            if ref.norm() < 1e-7:
                continue
            got = names[key[3:]].grad.cpu().flatten().double() / 1024.0
            rels.append(float((got - ref).norm() / ref.norm()))
            if dtype == "float16":
                cos = torch.dot(ref, got) / (ref.norm() * got.norm())
                assert cos > 0.999, (key, cos.item())
        errs[dtype] = float(np.median(rels))
    print("median rel-L2 gradient error on the tiny fixture: float16 %.2e, bfloat16 %.2e" % (errs["float16"], errs["bfloat16"]))
    assert errs["float16"] < errs["bfloat16"]
    model = _train_model(g, dtype="float16", p_drop=0.1)
    l1 = model(query=q, passage=p).loss
    l1.backward()
    assert torch.isfinite(l1) and all(torch.isfinite(p_.grad).all() for p_ in model.lm_q.parameters() if p_.grad is not None)

    # (2) the trainer's loss scaler
    model = _train_model(g)                                           # dtype comes from autocast under --fp16
    args = _trainer_args(tmp_path, fp16=True, fp16_init_scale=float(2 ** 40), fp16_growth_interval=4, max_steps=60, learning_rate=1e-3)
    trainer = DRTrainer(model=model, args=args)
    trainer.create_optimizer_and_scheduler(num_training_steps=60)
    w0 = model.lm_q.encoder.layer[0].output.dense.weight.detach().clone()
    first = float(trainer.training_step(model, (q, p)))
    trainer.optimizer_step()
    sc = trainer._loss_scaler()
    assert sc is not None and sc.skipped_steps() == 1 and float(sc.state[0]) == float(2 ** 39)      # 2^40 overflowed: skipped, halved
    assert torch.equal(model.lm_q.encoder.layer[0].output.dense.weight.detach(), w0)                  # nothing moved
    last = first
    for _ in range(59):
        last = float(trainer.training_step(model, (q, p)))
        trainer.optimizer_step()
    assert 1 <= sc.skipped_steps() < 40 and float(sc.state[0]) < 2 ** 30 and math.isfinite(last)
    assert last < first - 0.05, (first, last)
    assert not torch.equal(model.lm_q.encoder.layer[0].output.dense.weight.detach(), w0)
