"""A small reader of Rust `fn` signatures and `enum` definitions (test infrastructure): enough of the grammar for
`trait AccelProvider` (crates/runmat-accelerate-api/src/lib.rs) and for `impl AccelProvider for HipProvider` (shim/hip_provider.rs),
so that the two can be compared parameter by parameter without a Rust toolchain."""
import re


def _split_top(s: str, sep: str = ","):
    """Split at `sep` outside <>, (), []."""
    out, depth, cur = [], 0, ""
    i = 0
    while i < len(s):
        ch = s[i]
        if ch in "<([":
            depth += 1
        elif ch in ">)]":
            if ch == ">" and i > 0 and s[i - 1] == "-":  # the arrow of a closure type
                pass
            else:
                depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        out.append(cur)
    return [p.strip() for p in out]


def norm_type(t: str) -> str:
    t = re.sub(r"\s+", " ", t.strip())
    t = t.replace("anyhow::Result", "Result").replace("std::result::Result", "Result")
    t = re.sub(r"\bAccelDownloadFuture<'a>", "AccelProviderFuture<'a, HostTensorOwned>", t)
    t = re.sub(r"\s*,\s*", ", ", t)
    t = re.sub(r"\s*<\s*", "<", t)
    t = re.sub(r"\s*>", ">", t)
    return t


def split_ref(t: str):
    """('&\\'a' | '&' | '&mut' | '', base type)"""
    t = norm_type(t)
    m = re.match(r"&\s*('[a-z_]+)?\s*(mut\s+)?(.*)$", t)
    if not m:
        return "", t
    kind = "&" + (m.group(1) or "") + (" mut" if m.group(2) else "")
    return kind, m.group(3).strip()


def parse_signature(text: str):
    """`fn name<generics>(params) -> ret` (the text up to, not including, `{` / `;` / `where`)."""
    m = re.match(r"\s*(?:pub\s+)?(async\s+)?(?:unsafe\s+)?fn\s+([A-Za-z_0-9$]+)\s*(<[^(]*>)?\s*\(", text)
    if not m:
        return None
    name = m.group(2)
    i = m.end()
    depth, j = 1, i
    while j < len(text) and depth:
        depth += text[j] in "([" 
        depth -= text[j] in ")]"
        j += 1
    params_txt = text[i:j - 1]
    rest = text[j:]
    rm = re.match(r"\s*->\s*(.*)$", rest, flags=re.S)
    ret = norm_type(re.split(r"\bwhere\b", rm.group(1))[0]) if rm else "()"
    params = []
    for p in _split_top(params_txt):
        if not p:
            continue
        if re.match(r"&?\s*('[a-z_]+\s+)?(mut\s+)?self$", p):
            params.append({"name": "self", "ref": split_ref(p.replace("self", "Self"))[0], "type": "Self"})
            continue
        pm = re.match(r"(?:mut\s+)?([A-Za-z_0-9$]+)\s*:\s*(.*)$", p, flags=re.S)
        if not pm:
            params.append({"name": "?", "ref": "", "type": norm_type(p)})
            continue
        kind, base = split_ref(pm.group(2))
        params.append({"name": pm.group(1), "ref": kind, "type": base})
    return {"name": name, "async": bool(m.group(1)), "generics": norm_type(m.group(3) or ""), "params": params, "ret": ret}


def parse_fn_signatures(body: str, first_line: int = 1, indent: str = "    "):
    """Every `fn` declared at exactly `indent` inside `body` -> {name: signature + line}."""
    out = {}
    lines = body.splitlines()
    i = 0
    while i < len(lines):
        if re.match(rf"{indent}(?:async )?(?:unsafe )?fn [A-Za-z_0-9$]+", lines[i]):
            j, txt = i, ""
            while j < len(lines):
                txt += lines[j] + "\n"
                # the signature ends at the first `{` or `;` outside brackets
                flat = re.sub(r"\[[^\]]*;[^\]]*\]", "[]", txt)  # array types `[T; N]`
                cut = re.search(r"[{;]", flat)
                if cut:
                    txt = flat[:cut.start()]
                    break
                j += 1
            sig = parse_signature(txt)
            if sig:
                sig["line"] = first_line + i
                out[sig["name"]] = sig
            i = j + 1
        else:
            i += 1
    return out


def parse_enums(text: str):
    """{enum name: [variant names]} for every `pub enum` of the file."""
    out = {}
    for m in re.finditer(r"pub enum ([A-Za-z_0-9]+)(?:<[^>]*>)?\s*\{", text):
        depth, i = 1, m.end()
        while i < len(text) and depth:
            depth += text[i] == "{"
            depth -= text[i] == "}"
            i += 1
        body = text[m.end():i - 1]
        # variants: identifiers at nesting depth 0 that start an item
        variants, depth, item = [], 0, ""
        for ch in body + ",":
            if ch in "{(<[":
                depth += 1
            elif ch in "})>]":
                depth -= 1
            if ch == "," and depth == 0:
                item = re.sub(r"(?m)^\s*(//.*|#\[.*\])\s*$", "", item).strip()
                vm = re.match(r"([A-Z][A-Za-z_0-9]*)", item)
                if vm:
                    variants.append(vm.group(1))
                item = ""
            else:
                item += ch
        out[m.group(1)] = variants
    return out
