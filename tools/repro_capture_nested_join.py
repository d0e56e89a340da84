#!/usr/bin/env python3
"""DOCUMENTATION, not a test: the HIP-graph capture topology that aborts the process on ROCm 7.2 (gfx950)
and the ones that do not.  Pure torch, no libsept.  Run one mode per process on a GPU box:

    python tools/repro_capture_nested_join.py nested --i-know-this-dumps-core   # origin -> s1 -> wg, s1 joins wg    : core dump in
                                                         #   hipStreamEndCapture (round-1 logs cap_nested / cap_both
                                                         #   / cap_n4 / cap_n5: "the monitored command dumped core")
    python tools/repro_capture_nested_join.py origin     # origin -> s1 -> wg, ORIGIN joins wg : works (cap_n1)
    python tools/repro_capture_nested_join.py multifork  # origin -> wg1, origin -> wg2        : works (cap_multifork)
    python tools/repro_capture_nested_join.py sibling    # origin -> s1, origin -> s2, s1 waits for an event of s2,
                                                         #   both joined at the origin           : see DESIGN.md section 8

What sept_amd does about it: functional.fork_allowed() -- inside a capture a side stream is only forked when its
join lands on the capture's origin stream (published by capture_origin() in the trainers' capture()); any other
caller runs in line.  tests/test_model_gpu.py::test_capture_on_non_origin_stream_stays_in_line exercises the guard.
NEVER run the 'nested' mode inside a test run: it takes the process down."""
import sys

import torch


def main(mode):
    dev = torch.device("cuda")
    a = torch.randn(1 << 20, device=dev)
    s1, wg, wg2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        origin = torch.cuda.current_stream()
        if mode in ("nested", "origin"):
            s1.wait_stream(origin)
            with torch.cuda.stream(s1):
                b = a * 2
                wg.wait_stream(s1)
                with torch.cuda.stream(wg):
                    c = b + 1
                d = b - 1
                if mode == "nested":
                    s1.wait_stream(wg)        # a forked stream joins another forked stream: aborts at end of capture
            if mode == "origin":
                origin.wait_stream(wg)        # the same fork, joined at the origin: fine
            origin.wait_stream(s1)
            out = c + d
        elif mode == "sibling":
            s1.wait_stream(origin)
            wg.wait_stream(origin)
            with torch.cuda.stream(wg):
                c = a + 1
                ev = torch.cuda.Event()
                ev.record(wg)
                c2 = c * 3
            with torch.cuda.stream(s1):
                d = a - 1
                s1.wait_event(ev)             # a forked stream waits for a point of its SIBLING's chain
                e = c + d
            origin.wait_stream(wg)
            origin.wait_stream(s1)
            out = e + c2
            c = d = out
        else:
            wg.wait_stream(origin)
            wg2.wait_stream(origin)
            with torch.cuda.stream(wg):
                c = a + 1
            with torch.cuda.stream(wg2):
                d = a - 1
            origin.wait_stream(wg)
            origin.wait_stream(wg2)
            out = c + d
    g.replay()
    torch.cuda.synchronize()
    print(mode, "ok", float(out.sum()))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "origin"
    if mode == "nested" and "--i-know-this-dumps-core" not in sys.argv[2:]:
        sys.exit("the 'nested' mode aborts the process inside hipStreamEndCapture (core dump on the GPU box); "
                 "pass --i-know-this-dumps-core to run it anyway")
    main(mode)
