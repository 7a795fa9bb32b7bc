"""Feature ingest ("next" row 8f-3): replacement for the h5py-backed StartEndEvalDataset feature lookup + collate
(xml/start_end_dataset.py:208-232,297-359) on a box without h5py.

  FeatureStore        flat binary container: <path>.bin (all rows, float16 / float32, memory-mapped) + <path>.json
                      ({"dim", "dtype", "index": {name: [first_row, n_rows]}}).  `write_feature_store` converts a
                      {name: (n_clips, D) array} mapping (e.g. exported from the reference's h5 files elsewhere).
  ContextFeeder       iterator of (video_feat, video_mask, sub_feat, sub_mask) DEVICE batches with the reference's
                      semantics: truncate to max_ctx_len (start_end_dataset.py:311,320), pad with zeros to the batch
                      maximum + float mask (pad_sequences_1d), L2-normalise each clip x / (||x|| + 1e-5)
                      (utils/basic_utils.py:82-84) -- the normalisation runs on the device (xml_l2norm_rows_eps).
                      Rows are gathered straight from the memory map into pinned staging buffers (two, alternating)
                      and copied asynchronously on a side stream, so the H2D copy of batch i+1 overlaps the encoder
                      kernels of batch i.
  StoreEvalDataset    the reference's eval-dataset contract (set_data_mode / load_gt_vid_name_for_query / items with
                      "meta" + "model_inputs") over FeatureStores, for compute_context_info / compute_query2ctx_info.
"""
import json
import os

import numpy as np
import torch

from . import ops as hip_ops

_DT = {"float16": np.float16, "float32": np.float32}


def write_feature_store(path, features, dtype="float16"):
    names = list(features)
    dim = int(np.asarray(features[names[0]]).shape[1])
    index, row = {}, 0
    with open(path + ".bin", "wb") as f:
        for n in names:
            a = np.asarray(features[n])
            assert a.ndim == 2 and a.shape[1] == dim
            f.write(np.ascontiguousarray(a, dtype=_DT[dtype]).tobytes())
            index[n] = [row, int(a.shape[0])]
            row += int(a.shape[0])
    with open(path + ".json", "w") as f:
        json.dump(dict(dim=dim, dtype=dtype, rows=row, index=index), f)


class FeatureStore(object):
    def __init__(self, path):
        meta = json.load(open(path + ".json"))
        self.dim, self.index = meta["dim"], meta["index"]
        self.data = np.memmap(path + ".bin", dtype=_DT[meta["dtype"]], mode="r", shape=(meta["rows"], self.dim))

    def __contains__(self, name):
        return name in self.index

    def __getitem__(self, name):
        first, n = self.index[name]
        return self.data[first:first + n]

    def n_rows(self, name):
        return self.index[name][1]


class ContextFeeder(object):
    def __init__(self, video_names, video_store=None, sub_store=None, max_ctx_len=100, batch_size=200,
                 normalize_vfeat=True, normalize_tfeat=True, device="cuda:0", ops=hip_ops, feature_dtype=torch.float32):
        """feature_dtype=torch.bfloat16 (bf16 models only): the normalised features are handed over in bf16 -- the encoder's
        input LayerNorm reads half the bytes (the C ABI takes f32 or the compute dtype); the reference's contract is f32."""
        self.feature_dtype = feature_dtype
        self.names, self.vs, self.ss = list(video_names), video_store, sub_store
        self.max_ctx_len, self.bsz = int(max_ctx_len), int(batch_size)
        self.norm = dict(video=normalize_vfeat, sub=normalize_tfeat)
        self.device, self.ops = torch.device(device), ops
        self._stage = {}
        self._slot_done = {}     # (tag, slot) -> event recorded after the H2D copy that last read this pinned slot

    def __len__(self):
        return (len(self.names) + self.bsz - 1) // self.bsz

    def _staging(self, key, shape):
        buf = self._stage.get(key)
        if buf is None or buf.shape[0] < shape[0] or buf.shape[1] < shape[1]:
            buf = torch.zeros(shape, dtype=torch.float32)
            if self.device.type == "cuda":
                buf = buf.pin_memory()
            self._stage[key] = buf
        return buf[:shape[0], :shape[1]]

    def _gather(self, store, names, slot, tag):
        lens = [min(store.n_rows(n), self.max_ctx_len) for n in names]
        lmax = max(lens)
        ev = self._slot_done.get((tag, slot))
        if ev is not None:       # batch i - 2 was copied from this pinned slot asynchronously: the host must not zero /
            ev.synchronize()     # rewrite it before that copy has actually read it
        buf = self._staging((tag, slot), (len(names), lmax, store.dim))
        buf.zero_()
        mask = torch.zeros((len(names), lmax), dtype=torch.float32)
        out = buf.numpy()
        for i, (n, l) in enumerate(zip(names, lens)):
            out[i, :l] = store[n][:l]          # memmap -> pinned buffer, dtype-converting copy
            mask[i, :l] = 1
        return buf, mask

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
                host, mask = self._gather(store, names, bi & 1, tag)
                if cuda:
                    with torch.cuda.stream(copy_stream):
                        dev = host.to(self.device, non_blocking=True)
                        dmask = mask.to(self.device, non_blocking=True)
                        done = torch.cuda.Event()
                        done.record(copy_stream)
                    self._slot_done[(tag, bi & 1)] = done
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_stream(copy_stream)
                    dev.record_stream(cur)          # both were allocated on the copy stream and are consumed on the
                    dmask.record_stream(cur)        # compute stream: keep the allocator from recycling them early
                else:
                    dev, dmask = host.clone(), mask
                if self.norm[tag]:
                    dev = self.ops.l2norm_rows_eps(dev.contiguous(), 1e-5)
                if self.feature_dtype != torch.float32 and hasattr(self.ops, "convert"):
                    dev = self.ops.convert(dev.contiguous(), self.feature_dtype)
                out += [dev, dmask]
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
