#!/usr/bin/env python3
"""Walk the generator's backward chain at B = 16 / 256^2 on the MI355X: per layer, the HIP path's deviation from the fp64
oracle fixture next to the spread of three CPU fp32 evaluations (tests/golden/steps_chain.npz).  Environment switches of
the library (TG_NO_WINO, TG_NO_WINO44, TG_VGG_WINO4=0 ...) are read at first use, so an A/B is one process per setting:

    python tools/backward_chain.py --out gpurun_out/chain_default.json
    TG_NO_WINO=1 python tools/backward_chain.py --out gpurun_out/chain_nowino.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "terra-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("TERRAGAN_ALLOW_STANDIN_VGG", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--tag", default="c2_b16_256")
    ap.add_argument("--fixture", default="steps_chain")
    args = ap.parse_args()
    import torch
    from tests import chain_util as CU
    rows = CU.measure_chain(torch.device("cuda:0"), args.tag, fixture=args.fixture)
    print(CU.format_table(rows))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        env = {k: v for k, v in os.environ.items() if k.startswith("TG_")}
        with open(args.out, "w") as f:
            json.dump({"tag": args.tag, "env": env, "order": CU.ordered_keys(rows), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
