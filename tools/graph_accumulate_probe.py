"""Does the HIP runtime fault in hipGraphLaunch once enough forked hipGraphExec objects are ALIVE in one process?

Round 2's GPU suite segfaulted inside hipGraphLaunch (torch/cuda/graphs.py replay <- train.GraphedStep.__call__) after ~20
captured training steps had been built and -- caught in reference cycles -- never destroyed.  The runtime torch ships
(libamdhip64.so.7 of ROCm 7.0 inside torch/lib) gives every hipGraphExec its OWN parallel streams, one per extra branch of
the graph ("[hipGraph] Failed to create parallel stream!" is its message when that fails).  This probe builds ``--live``
graphs of ``--width`` parallel branches (main chain + width-1 event-forked side branches, the ops.SideStream pattern),
keeps every exec alive (or, with ``--close``, resets each graph after its replays) and replays each twice, then replays
all live ones again.  Each scenario runs in a child process so a fault in one cannot end the others; the parent never
touches the GPU.

  python tools/graph_accumulate_probe.py                       # the scenario table
  python tools/graph_accumulate_probe.py --child --live 64 --width 4 [--close]
"""
import argparse
import os
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--child", action="store_true")
ap.add_argument("--live", type=int, default=64)
ap.add_argument("--width", type=int, default=4)
ap.add_argument("--nodes", type=int, default=24)
ap.add_argument("--close", action="store_true")
ap.add_argument("--no-sync", action="store_true", help="with --close: reset without a device synchronize first")
ap.add_argument("--fresh-sides", action="store_true", help="new side streams for every capture instead of shared ones")
ap.add_argument("--defer", type=int, default=0, help="with --close: reset a graph only after this many newer ones exist")
ap.add_argument("--keep-graph", action="store_true", help="torch.cuda.CUDAGraph(keep_graph=True)")
ap.add_argument("--gdb", action="store_true")
a = ap.parse_args()

if not a.child:
    rows = []
    scenarios = [(1024, 4, []), (512, 4, ["--close"]), (512, 2, ["--close"]), (512, 1, ["--close"]),
                 (512, 4, ["--close", "--no-sync"]), (512, 4, ["--close", "--fresh-sides"]),
                 (512, 4, ["--close", "--defer", "8"]), (512, 4, ["--close", "--keep-graph"])]
    for live, width, extra in scenarios:
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--live", str(live), "--width", str(width)] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        last = [l for l in r.stdout.strip().splitlines() if l.startswith("probe")][-1:] or ["(no output)"]
        rows.append("live %4d width %d %-28s -> rc %4d  %s" % (live, width, " ".join(extra), r.returncode, last[0]))
        print(rows[-1], flush=True)
        if r.returncode != 0:
            print("   stderr tail: " + " | ".join(r.stderr.strip().splitlines()[-4:]), flush=True)
    sys.exit(0)

import faulthandler

import torch

faulthandler.enable()
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
ys = [torch.zeros(1 << 14, device=dev) for _ in range(a.width)]
sides = [torch.cuda.Stream(dev) for _ in range(a.width - 1)]      # shared by every capture, like ops._SIDE_STREAMS


def body():
    global sides
    if a.fresh_sides:
        sides = [torch.cuda.Stream(dev) for _ in range(a.width - 1)]
    main = torch.cuda.current_stream(dev)
    pending = []
    for i in range(a.nodes):
        x.add_(1.0)                                   # main chain first: it keeps the capture's own queue
        for ev, s, y in pending:
            s.wait_event(ev)
            with torch.cuda.stream(s):
                y.sin_()
        pending = []
        if i % 3 == 0:
            ev = torch.cuda.Event()
            ev.record(main)
            pending = [(ev, s, ys[j]) for j, s in enumerate(sides)]
    for ev, s, y in pending:
        s.wait_event(ev)
        with torch.cuda.stream(s):
            y.sin_()
    for s in sides:
        main.wait_stream(s)


s0 = torch.cuda.Stream(dev)
with torch.cuda.stream(s0):
    body()
torch.cuda.synchronize()
alive = []
for k in range(a.live):
    g = torch.cuda.CUDAGraph(keep_graph=True) if a.keep_graph else torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    g.replay()
    alive.append(g)
    if a.close and len(alive) > a.defer:
        if not a.no_sync:
            torch.cuda.synchronize()
        alive.pop(0).reset()
    if (k + 1) % 16 == 0:
        torch.cuda.synchronize()
        print("probe: %d graphs built, %d alive" % (k + 1, len(alive)), flush=True)
for g in alive:
    g.replay()
torch.cuda.synchronize()
print("probe ok: %d built, %d alive at the end, width %d, x[0]=%g" % (a.live, len(alive), a.width, float(x[0])), flush=True)
