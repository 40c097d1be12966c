"""Developer tool: rebuild the table in scripts/README.md from every script's first comment / docstring line."""
import re
from pathlib import Path

here = Path(__file__).resolve().parent
rows = []
for f in sorted(list(here.glob("*.py")) + list(here.glob("*.sh")) + list((here / "micro").glob("*.hip"))):
    text = f.read_text(errors="replace")
    desc = ""
    if f.suffix == ".py":
        m = re.match(r'(?:#![^\n]*\n)?\s*r?"""(.*?)"""', text, re.S)
        if m:
            desc = " ".join(m.group(1).split())
    if not desc:
        for line in text.splitlines():
            if line.startswith("#!"):
                continue
            if line.startswith("#") or line.startswith("//"):
                desc = line.lstrip("#/ ").strip()
                break
            if line.strip():
                break
    desc = desc.replace("|", "/")
    first = re.split(r"(?<=[.])\s", desc, maxsplit=1)[0] if desc else "(no header)"
    rows.append(f"| `{f.relative_to(here)}` | {first[:180]} |")
readme = here / "README.md"
head = readme.read_text().split("| script | what it measures |")[0]
readme.write_text(head + "| script | what it measures |\n|---|---|\n" + "\n".join(rows) + "\n")
print(f"{len(rows)} scripts indexed")
