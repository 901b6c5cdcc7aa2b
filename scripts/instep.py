"""Median duration of every mr:: kernel INSIDE training steps (bench.py's in_step_durations as a command).
Usage: python scripts/instep.py [--batch B] [--image-size S] [--image-height H]  -> one JSON object on stdout"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--batch", type=int, default=64)
p.add_argument("--image-size", type=int, default=256)
p.add_argument("--image-height", type=int, default=None)
a = p.parse_args()
print(json.dumps(bench.in_step_durations(a), indent=1, sort_keys=True))
