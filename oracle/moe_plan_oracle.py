"""CPU restatement (numpy, integer / index arithmetic) of the MoE routing plan - TEST INFRASTRUCTURE ONLY.

The reference has no such object: TimeMoeLayer / SpaceMoeLayer (Next-DiT-MoE/models/models2.py:459-506) loop over the experts on the
host - ``rows, nth = torch.where(selected == e)`` then ``results[rows] += weights[rows, nth] * expert(x[rows])`` - i.e. expert e
processes, IN ASCENDING ROW ORDER, the rows whose top-2 selection contains e.  The engine turns that loop into one grouped GEMM
over an expert-sorted copy of the rows; this file states what the sorted layout must be for the two to be the same computation:

  * entry i = 2 * row + k is the k-th (ascending expert id) selection of `row`;
  * expert e owns one contiguous segment of sorted positions that starts on a 256-row tile (segments in expert order, each padded
    up to a whole tile), and inside a segment the entries keep their order by i  (= ``torch.where``'s row order, :470-472);
  * pos[i] = sorted position of entry i; src[q] = row of the entry at sorted position q, -1 for padding positions;
    tile_expert[t] = the expert whose segment tile t lies in, -1 for tiles behind the last segment.

`route_time` restates the routing arithmetic itself for the time branch (:462-470): top-2 of the per-sample router logits, fp32
softmax over the two selected logits, weights cast to bf16, selections reported in ascending expert id.  EXACT TIES: the reference
calls torch.topk, whose pick among equal values is implementation-defined (the CPU build in this container returns experts (1, 3)
for logits (1, 1, 0.5, 1); CUDA's radix select is free to differ) - the engine and this restatement take the lowest indices.  In
fp32 (the parity fixtures) ties do not occur; with bf16-rounded logits they do, which is part of why the routing-agreement gate of
the full-depth MoE tests is a percentage and not an equality.
"""
import numpy as np

TILE = 256


def plan(sel: np.ndarray, n_experts: int, max_tiles: int):
    """sel int32 [rows, 2] -> (pos int32 [rows, 2], src int32 [max_tiles * 256], tile_expert int32 [max_tiles])"""
    flat = sel.reshape(-1)
    n = flat.shape[0]
    counts = np.array([(flat == e).sum() for e in range(n_experts)], dtype=np.int64)
    seg_tiles = (counts + TILE - 1) // TILE
    seg_off = np.concatenate([[0], np.cumsum(seg_tiles * TILE)])[:n_experts]
    assert int((seg_tiles).sum()) <= max_tiles, "sorted buffers too small"
    pos = np.empty(n, dtype=np.int32)
    src = np.full(max_tiles * TILE, -1, dtype=np.int32)
    tile_expert = np.full(max_tiles, -1, dtype=np.int32)
    for e in range(n_experts):
        idx = np.nonzero(flat == e)[0]  # ascending entry index = ascending row (an expert appears at most once per row)
        q = seg_off[e] + np.arange(idx.shape[0])
        pos[idx] = q
        src[q] = idx >> 1
        t0 = seg_off[e] // TILE
        tile_expert[t0: t0 + seg_tiles[e]] = e
    return pos.reshape(-1, 2), src, tile_expert


def _bf16_round(x: np.ndarray) -> np.ndarray:
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def route_time(logits_bf16: np.ndarray, rows_per_sample: int):
    """logits [B, E] (bf16-valued float32) -> (sel int32 [B * rows_per_sample, 2] ascending ids, wts float32 [.., 2] bf16-valued)"""
    B, E = logits_bf16.shape
    sel = np.empty((B, 2), dtype=np.int32)
    wts = np.empty((B, 2), dtype=np.float32)
    for b in range(B):
        l = logits_bf16[b].astype(np.float32)
        i1 = int(np.argmax(l))  # first maximum = lowest index
        rest = l.copy()
        rest[i1] = -np.inf
        i2 = int(np.argmax(rest))
        ex = np.exp(np.float32(l[i2] - l[i1]), dtype=np.float32)
        wa, wb = np.float32(1.0) / (np.float32(1.0) + ex), ex / (np.float32(1.0) + ex)
        if i2 < i1:
            sel[b], wts[b] = (i2, i1), (wb, wa)
        else:
            sel[b], wts[b] = (i1, i2), (wa, wb)
    wts = _bf16_round(wts)
    return np.repeat(sel, rows_per_sample, axis=0), np.repeat(wts, rows_per_sample, axis=0)
