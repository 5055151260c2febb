"""Token batches on the wire.  The reference's collators hand `input_ids`, `attention_mask` and `token_type_ids` to the
device as three int64 tensors (dataset/data_collator.py:78-91, retriever/dense_retriever.py:75): 3 KiB per 128-token
passage, pageable.  At the encoder's rate that is the copy engine's problem, not the GPU's, but it is 12x more than the
information in the batch: vocabularies fit 16 bits, masks of right-padded batches are a length per row, and token types
are all zero outside sentence-pair inputs.  `pack_token_batch` (called inside the collators, i.e. in the DataLoader
workers, so the result is what gets pinned) shrinks a batch to that; `unpack_token_batch` rebuilds the int64 tensors the
C ABI takes ON the device.  Anything that does not fit the compact form travels unchanged."""
import torch

PACKED_KEY = "_packed_tokens"


def pack_token_batch(batch):
    """dict of equal-shape integer tensors [B, L] -> compact dict (or the batch itself when it does not qualify)."""
    ids = batch.get("input_ids") if hasattr(batch, "get") else None
    if not torch.is_tensor(ids) or ids.dim() != 2 or ids.is_floating_point() or ids.numel() == 0:
        return batch
    lo, hi = int(ids.min()), int(ids.max())
    if lo < 0 or hi > 0xFFFF:
        return batch
    extra = set(batch.keys()) - {"input_ids", "attention_mask", "token_type_ids"}
    if extra:
        return batch
    out = {PACKED_KEY: torch.tensor([ids.shape[0], ids.shape[1]], dtype=torch.int64),
           "ids16": (ids.to(torch.int32) & 0xFFFF).to(torch.int16)}        # two's-complement carrier of the 16 bits
    mask = batch.get("attention_mask")
    if mask is not None:
        m = mask.to(torch.bool)
        prefix = bool((m[:, 1:] <= m[:, :-1]).all()) if m.shape[1] > 1 else True
        if prefix:
            out["lengths"] = m.sum(1).to(torch.int32)
        else:
            out["mask8"] = m.to(torch.uint8)
    tti = batch.get("token_type_ids")
    if tti is not None:
        if int(tti.max()) > 255 or int(tti.min()) < 0:
            return batch
        if bool((tti != 0).any()):
            out["tti8"] = tti.to(torch.uint8)
        else:
            out["tti_zero"] = torch.zeros(1, dtype=torch.uint8)
    return out


def is_packed(batch):
    return hasattr(batch, "keys") and PACKED_KEY in batch


def unpack_token_batch(batch, device, non_blocking=True):
    """The int64 tensors of the original batch, on `device` (packed or not)."""
    if not is_packed(batch):
        return {k: v.to(device, non_blocking=non_blocking) for k, v in batch.items()}
    n, L = (int(x) for x in batch[PACKED_KEY])
    dev = lambda t: t.to(device, non_blocking=non_blocking)
    out = {"input_ids": (dev(batch["ids16"]).to(torch.int32) & 0xFFFF).to(torch.int64)}
    if "lengths" in batch:
        lens = dev(batch["lengths"])
        out["attention_mask"] = (torch.arange(L, device=device, dtype=torch.int32)[None, :] < lens[:, None]).to(torch.int64)
    elif "mask8" in batch:
        out["attention_mask"] = dev(batch["mask8"]).to(torch.int64)
    if "tti8" in batch:
        out["token_type_ids"] = dev(batch["tti8"]).to(torch.int64)
    elif "tti_zero" in batch:
        out["token_type_ids"] = torch.zeros(n, L, dtype=torch.int64, device=device)
    return out


def model_batch(batch, device, model):
    """What the encode loops hand to `model`: the compact batch itself when the model widens it on its own (this package's
    DRModel / RRModel: `accepts_compact_batches` -- they read the host-side lengths first, which is what lets the encoder
    skip the padding rows, then call unpack_token_batch), the int64 tensors on `device` for any other model."""
    if is_packed(batch) and getattr(model, "accepts_compact_batches", False):
        return batch
    return unpack_token_batch(batch, device)


def token_rows_bound(batch):
    """Row bound of om_encoder_forward_packed (include/openmatch_hip.h) for a compact batch whose `lengths` still live on
    the host: the sequences' token counts (a row without any token counts as L, as the encoder treats it), summed and
    rounded up to whole 256-row tiles.  None when the batch carries no host-side lengths -- the padded entry runs then."""
    if not is_packed(batch) or "lengths" not in batch or batch["lengths"].is_cuda:
        return None
    n, L = (int(x) for x in batch[PACKED_KEY])
    lens = batch["lengths"].to(torch.int64)
    total = int(torch.where(lens > 0, lens, torch.full_like(lens, L)).sum())
    return (total + 255) // 256 * 256


def packed_nbytes(batch):
    return sum(v.numel() * v.element_size() for v in batch.values() if torch.is_tensor(v))
