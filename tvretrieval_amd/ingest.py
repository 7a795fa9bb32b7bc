"""Feature ingest ("next" row 8f-3): replacement for the h5py-backed StartEndEvalDataset feature lookup + collate
(xml/start_end_dataset.py:208-232,297-359) on a box without h5py.

  FeatureStore        flat binary container: <path>.bin (all rows, float16 / float32, memory-mapped) + <path>.json
                      ({"dim", "dtype", "index": {name: [first_row, n_rows]}}).  `write_feature_store` converts a
                      {name: (n_clips, D) array} mapping (e.g. exported from the reference's h5 files elsewhere).
  ContextFeeder       iterator of (video_feat, video_mask, sub_feat, sub_mask) DEVICE batches with the reference's
                      semantics: truncate to max_ctx_len (start_end_dataset.py:311,320), pad with zeros to the batch
                      maximum + float mask (pad_sequences_1d), L2-normalise each clip x / (||x|| + 1e-5)
                      (utils/basic_utils.py:82-84).  The host only moves bytes: raw rows go from the memory map into
                      pinned staging buffers (two, alternating) in the store's dtype and back to back, are copied
                      asynchronously on a side stream (the H2D copy of batch i+1 overlaps the encoder kernels of batch
                      i), and ONE device launch (xml_ingest_rows) truncates, pads, converts, normalises and writes
                      the mask.
  StoreEvalDataset    the reference's eval-dataset contract (set_data_mode / load_gt_vid_name_for_query / items with
                      "meta" + "model_inputs") over FeatureStores, for compute_context_info / compute_query2ctx_info.
"""
import json
import os

import numpy as np
import torch

from . import ops as hip_ops

_DT = {"float16": np.float16, "float32": np.float32}


class FeatureStoreWriter(object):
    """Streaming writer of a FeatureStore: add(name, (n_clips, D) array) one video at a time, close() writes the index."""

    def __init__(self, path, dim, dtype="float16"):
        self.path, self.dim, self.dtype = path, int(dim), dtype
        self.index, self.row = {}, 0
        self.f = open(path + ".bin", "wb")

    def add(self, name, a):
        a = np.asarray(a)
        assert a.ndim == 2 and a.shape[1] == self.dim
        self.f.write(np.ascontiguousarray(a, dtype=_DT[self.dtype]).tobytes())
        self.index[name] = [self.row, int(a.shape[0])]
        self.row += int(a.shape[0])

    def add_block(self, names, block):
        """Several videos of equal length at once: block (len(names), n_clips, D) -- one write."""
        block = np.ascontiguousarray(block, dtype=_DT[self.dtype])
        assert block.ndim == 3 and block.shape[0] == len(names) and block.shape[2] == self.dim
        self.f.write(memoryview(block).cast("B"))
        for n in names:
            self.index[n] = [self.row, int(block.shape[1])]
            self.row += int(block.shape[1])

    def close(self):
        self.f.close()
        with open(self.path + ".json", "w") as f:
            json.dump(dict(dim=self.dim, dtype=self.dtype, rows=self.row, index=self.index), f)


def write_feature_store(path, features, dtype="float16"):
    names = list(features)
    w = FeatureStoreWriter(path, int(np.asarray(features[names[0]]).shape[1]), dtype)
    for n in names:
        w.add(n, features[n])
    w.close()


class FeatureStore(object):
    def __init__(self, path):
        meta = json.load(open(path + ".json"))
        self.dim, self.index, self.dtype = meta["dim"], meta["index"], meta["dtype"]
        self.data = np.memmap(path + ".bin", dtype=_DT[meta["dtype"]], mode="r", shape=(meta["rows"], self.dim))

    def __contains__(self, name):
        return name in self.index

    def __getitem__(self, name):
        first, n = self.index[name]
        return self.data[first:first + n]

    def n_rows(self, name):
        return self.index[name][1]


_TORCH_DT = {"float16": torch.float16, "float32": torch.float32}


class ContextFeeder(object):
    """Raw clip rows leave the host in the STORE's dtype (f16 on disk: half the PCIe bytes of the reference's f32 batches)
    and back to back -- runs of videos that are adjacent in the store are ONE copy from the memory map into the pinned
    staging buffer, spread over `host_threads` threads --; truncation, padding, the mask, the conversion and the per-clip
    normalisation happen in one device launch (xml_ingest_rows).  Two staging buffers alternate, the copy runs on a side
    stream: the H2D of batch i + 1 overlaps the encoder kernels of batch i.
    stats (after iterating): rows / bytes moved and the seconds the host spent gathering."""

    def __init__(self, video_names, video_store=None, sub_store=None, max_ctx_len=100, batch_size=200,
                 normalize_vfeat=True, normalize_tfeat=True, device="cuda:0", ops=hip_ops, feature_dtype=torch.float32,
                 host_threads=8):
        """feature_dtype=torch.bfloat16 (bf16 models only): the normalised features are handed over in bf16 -- the encoder's
        input LayerNorm reads half the bytes (the C ABI takes f32 or the compute dtype); the reference's contract is f32."""
        self.feature_dtype = feature_dtype
        self.names, self.vs, self.ss = list(video_names), video_store, sub_store
        self.max_ctx_len, self.bsz = int(max_ctx_len), int(batch_size)
        self.norm = dict(video=normalize_vfeat, sub=normalize_tfeat)
        self.device, self.ops = torch.device(device), ops
        self._stage = {}
        self._slot_done = {}     # (tag, slot) -> event recorded after the H2D copy that last read this pinned slot
        self._pool = None
        self.host_threads = max(1, int(host_threads))
        self.stats = dict(rows=0, h2d_bytes=0, gather_s=0.0)

    def __len__(self):
        return (len(self.names) + self.bsz - 1) // self.bsz

    def _staging(self, key, rows, store):
        buf = self._stage.get(key)
        if buf is None or buf.shape[0] < rows:
            cap = max(rows, min(self.bsz, len(self.names)) * self.max_ctx_len)
            buf = torch.empty((cap, store.dim), dtype=_TORCH_DT[store.dtype])
            if self.device.type == "cuda":
                buf = buf.pin_memory()
            self._stage[key] = buf
        return buf

    def _gather(self, store, names, slot, tag):
        """-> (pinned (rows, D) buffer in the store dtype holding the batch's truncated videos back to back, rows,
        row_start (n + 1) int64, lmax)"""
        import time
        t0 = time.perf_counter()
        idx = [store.index[n] for n in names]
        lens = np.minimum(np.array([i[1] for i in idx], dtype=np.int64), self.max_ctx_len)
        start = np.concatenate([[0], np.cumsum(lens)])
        rows = int(start[-1])
        ev = self._slot_done.get((tag, slot))
        if ev is not None:       # batch i - 2 was copied from this pinned slot asynchronously: the host must not rewrite it
            ev.synchronize()     # before that copy has actually read it
        buf = self._staging((tag, slot), rows, store)
        out = buf.numpy()
        # runs: consecutive videos that are adjacent in the store and taken whole collapse into one copy
        runs, i = [], 0
        while i < len(idx):
            src, dst, n = idx[i][0], int(start[i]), int(lens[i])
            while i + 1 < len(idx) and lens[i] == idx[i][1] and idx[i + 1][0] == idx[i][0] + idx[i][1]:
                i += 1
                n += int(lens[i])
            runs.append((src, dst, n))
            i += 1
        chunk = max(1, (8 << 20) // (store.dim * out.itemsize))              # ~8 MB pieces
        jobs = [(s + o, d + o, min(chunk, n - o)) for s, d, n in runs for o in range(0, n, chunk)]

        def copy(j):
            out[j[1]:j[1] + j[2]] = store.data[j[0]:j[0] + j[2]]             # (numpy releases the GIL inside the copy loop)
        if len(jobs) > 1 and self.host_threads > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.host_threads)
            list(self._pool.map(copy, jobs))
        else:
            for j in jobs:
                copy(j)
        self.stats["gather_s"] += time.perf_counter() - t0
        return buf, rows, torch.from_numpy(start), int(lens.max())

    def __iter__(self):
        cuda = self.device.type == "cuda"
        copy_stream = torch.cuda.Stream(self.device) if cuda else None
        for bi, b in enumerate(range(0, len(self.names), self.bsz)):
            names = self.names[b:b + self.bsz]
            out = []
            for tag, store in (("video", self.vs), ("sub", self.ss)):
                if store is None:
                    out += [None, None]
                    continue
                host, rows, start, lmax = self._gather(store, names, bi & 1, tag)
                self.stats["rows"] += rows
                self.stats["h2d_bytes"] += rows * store.dim * host.element_size()
                if cuda:
                    with torch.cuda.stream(copy_stream):
                        dev = host[:rows].to(self.device, non_blocking=True)
                        dstart = start.to(self.device, non_blocking=True)
                        done = torch.cuda.Event()
                        done.record(copy_stream)
                    self._slot_done[(tag, bi & 1)] = done
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_stream(copy_stream)
                    dev.record_stream(cur)          # both were allocated on the copy stream and are consumed on the
                    dstart.record_stream(cur)       # compute stream: keep the allocator from recycling them early
                else:
                    dev, dstart = host[:rows].clone(), start
                feat, mask = self.ops.ingest_rows(dev, dstart, len(names), lmax, self.max_ctx_len, normalize=self.norm[tag],
                                                  eps=1e-5, out_dtype=self.feature_dtype)
                out += [feat, mask]
            yield tuple(out)


class StoreEvalDataset(object):
    """Reference eval-dataset contract over FeatureStores.  query_data: list of dicts with desc_id, desc, vid_name
    (+ ts/type for evaluation); video_data: list of dicts with vid_name, duration; video2idx: name -> int."""

    def __init__(self, query_data, video_data, video2idx, desc_store, video_store=None, sub_store=None, max_desc_len=30,
                 max_ctx_len=100, normalize_vfeat=True, normalize_tfeat=True):
        self.query_data, self.video_data, self.video2idx = query_data, video_data, video2idx
        self.desc, self.vs, self.ss = desc_store, video_store, sub_store
        self.max_desc_len, self.max_ctx_len = max_desc_len, max_ctx_len
        self.nv, self.nt = normalize_vfeat, normalize_tfeat
        self.data_mode, self.load_gt_video = "query", False

    def set_data_mode(self, mode):
        assert mode in ("context", "query")
        self.data_mode = mode

    def load_gt_vid_name_for_query(self, flag):
        self.load_gt_video = flag

    def __len__(self):
        return len(self.query_data) if self.data_mode == "query" else len(self.video_data)

    @staticmethod
    def _norm(a):
        a = np.asarray(a, dtype=np.float32)
        return a / (np.linalg.norm(a, axis=-1, keepdims=True) + 1e-5)

    def __getitem__(self, i):
        if self.data_mode == "context":
            v = self.video_data[i]
            mi = {}
            if self.vs is not None:
                f = self.vs[v["vid_name"]][:self.max_ctx_len]
                mi["video_feat"] = self._norm(f) if self.nv else np.asarray(f, dtype=np.float32)
            if self.ss is not None:
                f = self.ss[v["vid_name"]][:self.max_ctx_len]
                mi["sub_feat"] = self._norm(f) if self.nt else np.asarray(f, dtype=np.float32)
            return dict(meta=dict(vid_name=v["vid_name"], duration=v.get("duration", 0.0)), model_inputs=mi)
        q = self.query_data[i]
        f = self.desc[str(q["desc_id"])][:self.max_desc_len]
        meta = dict(desc_id=q["desc_id"], desc=q.get("desc", ""),
                    vid_name=q["vid_name"] if self.load_gt_video else None)
        return dict(meta=meta, model_inputs=dict(query_feat=self._norm(f) if self.nt else np.asarray(f, dtype=np.float32)))
