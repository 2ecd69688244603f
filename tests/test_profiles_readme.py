"""profiles/README.md quotes the committed rocprofv3 / PMC / bench summaries.  Its numbers block is GENERATED from those
files (tools/profiles_readme.py); this test regenerates it and compares, so the index and the JSON / CSV cannot disagree
(a hand-written row said wino_wgrad busy 0.47 and all-matrix 0.48 where the committed JSON says 0.277 / 0.444)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_block_of_the_profiles_readme_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profiles_readme.py"), "--check"], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr


def test_readme_numbers_equal_the_committed_json():
    txt = open(os.path.join(ROOT, "profiles", "README.md")).read()
    for tag in sorted({m for m in re.findall(r"### (r\d\d): `r\d\d_pmc_f32_mfma.json`", txt)}):
        d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_f32_mfma.json")))
        block = txt[txt.index(f"### {tag}: `{tag}_pmc_f32_mfma.json`"):]
        block = block[:block.index("\n### ", 5)] if "\n### " in block[5:] else block
        for fam, v in d["families"].items():
            row = next(ln for ln in block.splitlines() if ln.startswith(f"| {fam} |"))
            assert f"| {v['mfma_pipe_busy']:.3f} |" in row, (tag, fam, row)
        assert f"**{d['all_matrix_kernels']['mfma_pipe_busy']:.3f}**" in block
    # the two figures the round-3 index had wrong
    d3 = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_f32_mfma.json")))
    ww = next(v for k, v in d3["families"].items() if k.startswith("wino_wgrad"))
    assert abs(ww["mfma_pipe_busy"] - 0.277) < 5e-4 and abs(d3["all_matrix_kernels"]["mfma_pipe_busy"] - 0.444) < 5e-4
    assert "busy 0.47" not in txt and "all matrix kernels 0.48" not in txt
