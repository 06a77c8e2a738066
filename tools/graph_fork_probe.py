"""How the HIP runtime runs a forked hipGraph: a main chain of N kernels, after every `every`-th of them a fork onto a
side stream (event record / wait, the pattern of ops.SideStream) with one side kernel, joined at the end.  Run under
`rocprofv3 --kernel-trace` and reduce with tools/graph_fork_probe.py --reduce <dir>: when does each side kernel start
relative to the main kernel it depends on?
  python tools/graph_fork_probe.py [--n 200] [--every 2] [--main-elems 67108864] [--side-elems 1048576] [--side-first 0]"""
import argparse, csv, glob, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200); ap.add_argument("--every", type=int, default=2)
ap.add_argument("--main-elems", type=int, default=1 << 26); ap.add_argument("--side-elems", type=int, default=1 << 20)
ap.add_argument("--side-first", type=int, default=0, help="1: capture the side kernel before the next main kernel at a fork")
ap.add_argument("--reduce", default=None)
ap.add_argument("--side-kind", default="sin", help="sin: elementwise over --side-elems; mm: a [256 x K] x [K x 256] GEMM (few workgroups, long)")
ap.add_argument("--side-k", type=int, default=16384)
a = ap.parse_args()
if a.reduce:
    f = glob.glob(a.reduce + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    is_side = lambda r: 'sin' in r['Kernel_Name']
    is_main = lambda r: 'CUDAFunctor_add' in r['Kernel_Name'] or 'AddFunctor' in r['Kernel_Name'] or ('add' in r['Kernel_Name'].lower() and 'sin' not in r['Kernel_Name'])
    mains = [r for r in rows if is_main(r)]; sides = [r for r in rows if is_side(r)]
    n_main = int(os.environ.get("PROBE_N", "200")); every = int(os.environ.get("PROBE_EVERY", "2"))
    mains = mains[-n_main:]; sides = sides[-(n_main // every):]      # last replay
    t0 = int(mains[0]['Start_Timestamp'])
    print("main: %d kernels %.3f -> %.3f ms on queues %s" % (len(mains), 0.0, (int(mains[-1]['End_Timestamp']) - t0) / 1e6,
                                                            sorted(set(r['Queue_Id'] for r in mains))))
    print("side: %d kernels on queues %s" % (len(sides), sorted(set(r['Queue_Id'] for r in sides))))
    for j, r in enumerate(sides):
        dep = mains[(j + 1) * every - 1]
        lag = (int(r['Start_Timestamp']) - int(dep['End_Timestamp'])) / 1e3
        if j < 6 or j % 10 == 0 or j == len(sides) - 1:
            print("  side %3d: dependency (main %3d) ends %8.3f ms, side starts %8.3f ms  (lag %8.1f us)" % (
                j, (j + 1) * every - 1, (int(dep['End_Timestamp']) - t0) / 1e6, (int(r['Start_Timestamp']) - t0) / 1e6, lag))
    sys.exit(0)
import torch
dev = "cuda:0"
x = torch.zeros(a.main_elems, device=dev); y = torch.zeros(a.side_elems, device=dev)
ma = torch.randn(256, a.side_k, device=dev, dtype=torch.bfloat16); mb = torch.randn(a.side_k, 256, device=dev, dtype=torch.bfloat16)
mc = torch.empty(256, 256, device=dev, dtype=torch.bfloat16)
def side_op():
    if a.side_kind == "mm":
        torch.mm(ma, mb, out=mc)
    else:
        torch.sin_(y)
side = torch.cuda.Stream(dev)
def step():
    main = torch.cuda.current_stream(dev)
    for i in range(a.n):
        x.add_(1.0)
        if (i + 1) % a.every == 0:
            ev = torch.cuda.Event(); ev.record(main)
            if a.side_first:
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    side_op()
                if i + 1 < a.n:
                    pass
            else:
                # the next main kernel is captured first (below, next iteration); remember the event
                pending.append(ev)
        if not a.side_first and pending and (i + 1) % a.every == 1 % a.every and i > 0:
            ev = pending.pop(0)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                side_op()
    while pending:
        ev = pending.pop(0); side.wait_event(ev)
        with torch.cuda.stream(side):
            side_op()
    main.wait_stream(side)
pending = []
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    import time
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    # the parts alone, eagerly, for reference
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(a.n):
        x.add_(1.0)
    torch.cuda.synchronize(); t_main = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for i in range(a.n // a.every):
        side_op()
    torch.cuda.synchronize(); t_side = (time.perf_counter() - t0) * 1e3
print("graph replay %.2f ms (min of 5);  main chain alone %.2f ms, side kernels alone %.2f ms" % (min(ts), t_main, t_side))
