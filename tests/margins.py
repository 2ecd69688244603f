"""Measured parity margins on file: when $DV_PARITY_MARGINS names a file, the whole-step oracle tests
append one line per compared tensor - the engine's distance from the oracle, the distance of an independent float32 (or bf16)
evaluation of the same formulas where the test computes one, and the bound the assertion used - so that the rule
"1e-3, or 1.5 x float32's own distance where that is larger, never above 1e-2" (tests/test_gpu_parity.py::_run_parity) can be
audited from a tracked file (profiles/rNN_parity_margins.txt) instead of a scratch log."""
import os


def record(case, rows, header=None):
    """rows: iterable of (tensor, engine_err, reference_impl_err or None, bound, note)."""
    path = os.environ.get("DV_PARITY_MARGINS")
    if not path:
        return
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    with open(path, "a") as fh:
        fh.write(f"# {case}\n")
        if header:
            fh.write(f"#   {header}\n")
        fh.write(f"#   {'tensor':34s} {'engine':>10s} {'other impl':>10s} {'bound':>10s}  note\n")
        for name, e, f, b, note in rows:
            fs = f"{f:10.3e}" if f is not None else f"{'-':>10s}"
            fh.write(f"    {name:34s} {e:10.3e} {fs} {b:10.3e}  {note}\n")
