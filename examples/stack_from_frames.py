#!/usr/bin/env python3
"""The reference's `examples/stack-from-frames.fsp` project (align -> balance -> focus stack) as a
script on the MI355X path.  Same action graph, same parameter names as shinestacker's
StackJob / CombinedActions / AlignFrames / BalanceFrames / FocusStack; only the imports differ.

    python examples/stack_from_frames.py <working_dir> <input_subdir> [--no-align]

Without OpenCV the transform is estimated by the GPU ECC estimator (shinestacker_amd.align.ecc_estimator);
with OpenCV installed, drop the `estimator=` argument to use the reference's SIFT + RANSAC recipe."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from shinestacker_amd import (AlignFrames, BalanceFrames, CombinedActions, FocusStack, PyramidStack,  # noqa: E402
                              StackJob)
from shinestacker_amd.align import ecc_estimator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("working_dir")
    ap.add_argument("input_subdir")
    ap.add_argument("--no-align", action="store_true")
    args = ap.parse_args()
    job = StackJob("focus-stack", args.working_dir, input_path=args.input_subdir)
    stack_input = args.input_subdir
    if not args.no_align:
        job.add_action(CombinedActions("align-and-balance",
                                       [AlignFrames(estimator=ecc_estimator()),
                                        BalanceFrames(channel="RGB", corr_map="MATCH_HIST")],
                                       output_path="align"))
        stack_input = "align"
    job.add_action(FocusStack("stack", PyramidStack(), input_path=stack_input, output_path="stack",
                              prefix="stack_"))
    job.run()
    print("written:", sorted(os.listdir(os.path.join(args.working_dir, "stack"))))


if __name__ == "__main__":
    main()
