"""Extracts the few results the reference notebook pins into a JSON fixture.

Run in the build container only (the reference tree does not travel to the GPU box):
    python tests/golden/make_reference_fixture.py
Reads /root/reference/test.ipynb as JSON (nothing is executed) and writes
tests/golden/reference_notebook_fixture.json.  The fixture holds data only: index lists
printed by two notebook cells plus a handful of configuration values.
"""
import json
import re
from pathlib import Path

NOTEBOOK = Path("/root/reference/test.ipynb")
OUT = Path(__file__).resolve().parent / "reference_notebook_fixture.json"

CELLS = {
    "labels['heatmaps'][0].nonzero()": "heatmaps_nonzero",
    "labels['vecmaps'][0].nonzero()": "vecmaps_nonzero",
}


def main() -> None:
    nb = json.loads(NOTEBOOK.read_text())
    fix = {
        "_source": (
            "Outputs printed by the reference notebook test.ipynb: raw-JSON lines 606-614 "
            "(labels['heatmaps'][0].nonzero()) and 634-653 (labels['vecmaps'][0].nonzero()); "
            "index lists of the non-zero cells of one 40x40x3 heat-map label (NHWC) and of its "
            "40x40x8 vector-map label; and the (x, y) offsets the drawing cell prints per sample and "
            "pair over the first joint's cells (stream output of the matplotlib cell).  Config "
            "values from lines 919, 928, 1001-1010, 423, 453."
        ),
        "heatmap_shape_hwc": [40, 40, 3],
        "stride": 8,
        "input_hw": [320, 320],
        "adam": {"beta1": 0.9, "beta2": 0.999, "epsilon": "1e-8", "learning_rate": 0.001,
                 "decay_rate": 0.95, "decay_steps": 10000},
        "anchors": {"count": 1500, "grids": [[20, 20, 3], [10, 10, 3]]},
    }
    for cell in nb["cells"]:
        src = "".join(cell.get("source", [])).strip()
        if src in CELLS:
            txt = "".join(cell["outputs"][0]["data"]["text/plain"])
            arrs = re.findall(r"array\(\[(.*?)\]\)", txt, flags=re.S)
            rows, cols, chans = ([int(x) for x in a.replace("\n", " ").split(",")] for a in arrs)
            fix[CELLS[src]] = {"rows": rows, "cols": cols, "channels": chans}
        if src.startswith("import matplotlib.pyplot as plt") and "vecmaps[y, x, 4 * i]" in src:
            # the drawing cell prints, per sample and per pair, (vecmaps[y,x,4i], vecmaps[y,x,4i+1])
            # over the first joint's non-zero heat-map cells in row-major order
            txt = "".join(o.get("text", "") if isinstance(o.get("text"), str) else "".join(o.get("text", []))
                          for o in cell["outputs"] if o.get("output_type") == "stream")
            groups, cur = [], None
            for line in txt.splitlines():
                m = re.match(r"^(\d+) (\d+) \*+$", line.strip())
                if m:
                    cur = {"pair": [int(m.group(1)), int(m.group(2))], "values": []}
                    groups.append(cur)
                elif line.strip() and cur is not None:
                    a, b = line.split()
                    cur["values"].append([float(a), float(b)])
            fix["vecmap_printed_values"] = groups
    OUT.write_text(json.dumps(fix, separators=(",", ":")) + "\n")
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main()
