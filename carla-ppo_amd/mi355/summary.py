"""Minimal summary writer standing in for tf.summary.FileWriter (tensorboard is not installed here).

Writes one JSON object per line to <log_dir>/events.jsonl: {"tag", "value", "step", "wall_time"}.  It is never on a
timed path.  The TensorBoard-compatible TFRecord writer is a "next" item.
"""
import json
import os
import time


class SummaryWriter:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "events.jsonl")
        self._f = open(self.path, "a")

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step), "wall_time": time.time()}) + "\n")

    def add_text(self, tag, text, step=0):
        self._f.write(json.dumps({"tag": tag, "text": text, "step": int(step), "wall_time": time.time()}) + "\n")

    def flush(self):
        self._f.flush()

    def close(self):
        try:
            self._f.close()
        except Exception:
            pass
