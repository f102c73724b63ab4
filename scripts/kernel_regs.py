"""Register / scratch / LDS budget of every kernel in the device assembly (tools/_gen/render_gfx950.s by default)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "tools/_gen/render_gfx950.s"
pat = sys.argv[2] if len(sys.argv) > 2 else ""
name = None
rows = {}
for line in open(path, errors="replace"):
    m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
    if m:
        name = m.group(1)
        rows[name] = {}
        continue
    if name:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size|group_segment_fixed_size|accum_offset)\s+(\S+)", line)
        if m:
            rows[name][m.group(1)] = m.group(2)
        if ".end_amdhsa_kernel" in line:
            name = None
# spill counts live in the comment block after each function
spills = {}
cur = None
for line in open(path, errors="replace"):
    m = re.match(r"^(_Z\S+):", line)
    if m:
        cur = m.group(1)
    m = re.match(r";\s*(ScratchSize|VGPRSpill|NumVgprs|Occupancy|codeLenInByte)\s*[:=]?\s*(\d+)", line.replace("[", " ").replace("]", " "))
    if m and cur:
        spills.setdefault(cur, {})[m.group(1)] = m.group(2)
import subprocess
for k, v in rows.items():
    try:
        dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = k
    if pat and pat not in dem:
        continue
    s = spills.get(k, {})
    print("%-90s vgpr %3s sgpr %3s scratch %4s spill %3s occ %s code %s" % (dem[:90], v.get("next_free_vgpr"), v.get("next_free_sgpr"),
          v.get("private_segment_fixed_size"), s.get("VGPRSpill", "?"), s.get("Occupancy", "?"), s.get("codeLenInByte", "?")))
