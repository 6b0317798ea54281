#!/usr/bin/env python3
"""Parity at TRAINED operating points (``bench.trained_parity``; also ``python bench.py --parity-trained N``).

    python profiles/scripts/parity_trained.py [--sets 4] [--steps 3000] [--size 512] [--model hsic]
Prints one JSON line per (weight set, mode)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--lmbda", type=float, default=0.0067)
    ap.add_argument("--model", choices=["hsic", "joint"], default="hsic")
    args = ap.parse_args()
    import bench
    for r in bench.trained_parity(args.model, args.sets, args.steps, args.size, lmbda=args.lmbda, log=lambda t: print(t, file=sys.stderr, flush=True)):
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
