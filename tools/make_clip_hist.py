"""Clip-count histogram of the real TVR corpus (all 21 793 videos of data/tvr_video2dur_idx.json): ceil(duration / 1.5 s)
clipped to 128 -- the "ragged variant with lengths drawn from the real duration table" of SURVEY.md 8(d).  The table itself
stays in the reference tree; what is committed (tests/golden/tvr_clip_count_hist.json) is this 129-bin histogram, data
derived from a data file.  Run in the dev container:  python tools/make_clip_hist.py"""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/data/tvr_video2dur_idx.json"
CLIP = 1.5          # ProposalConfigs["tvr"]["clip_length"], xml/config.py:7,213


def main():
    table = json.load(open(SRC))
    hist = [0] * 129
    n = 0
    raw_max = 0
    for split, vids in table.items():
        for name, (dur, idx) in vids.items():
            c = int(math.ceil(dur / CLIP))
            raw_max = max(raw_max, c)
            hist[min(max(c, 1), 128)] += 1
            n += 1
    mean = sum(i * h for i, h in enumerate(hist)) / n
    out = dict(source="data/tvr_video2dur_idx.json (all splits)", clip_length=CLIP, n_videos=n, clipped_at=128,
               max_unclipped=raw_max, mean_clips=mean, hist=hist)
    path = os.path.join(ROOT, "tests", "golden", "tvr_clip_count_hist.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, "videos", n, "mean clips %.2f" % mean, "max unclipped", raw_max)
    b = [sum(hist[:33]), sum(hist[33:65]), sum(hist[65:])]
    print("buckets <=32 / <=64 / <=128:", b, "padded mean", (32 * b[0] + 64 * b[1] + 128 * b[2]) / n)


if __name__ == "__main__":
    main()
