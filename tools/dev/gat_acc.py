import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import train_res, odeint as OI
import io, contextlib
def run(argv, capture=True):
    old = OI.GRAPH_CAPTURE_MAX_ELEMS
    OI.GRAPH_CAPTURE_MAX_ELEMS = old if capture else 0
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            train_res.main(argv)
    finally:
        OI.GRAPH_CAPTURE_MAX_ELEMS = old
    out = [l for l in buf.getvalue().splitlines() if l.startswith("Test set results: loss") or l.startswith("Epoch: 0200") or l.startswith("Epoch: 0001")]
    print(" ".join(argv), "| capture", capture, "|", " || ".join(o[:110] for o in out), flush=True)
base = ["--variant", "gat", "--dataset", "cora"]
run(base + ["--model", "gcn3"])
run(base + ["--model", "res3"])
run(base + ["--model", "ode3", "--method", "rk4", "--step_size", "0.25"])
run(base + ["--model", "ode3", "--method", "rk4", "--step_size", "0.25"], capture=False)
