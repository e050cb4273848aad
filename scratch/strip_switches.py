"""Resolve a fixed set of preprocessor switches in a header (a small unifdef): python scratch/strip_switches.py in.h out.h
UNDEF: macros treated as undefined; VALUE: macros with a fixed integer value.  Every other conditional is left alone."""
import re, sys
UNDEF = {"CI_EXP_PRIO_INIT", "CI_EXP_PRIO_TRSM", "CI_EXP_FP32_EMUL", "CI_EXP_ALIAS", "CI_EXP_NOINIT", "CI_EXP_AHEAD_EMUL", "CI_EXP_NOFACTOR",
         "CI_EXP_NOTRSM", "CI_EXP_NOSTORE", "CI_EXP_STAGGER", "LA_EXP_NOINV", "MEDGP_LEGACY_AB"}
VALUE = {"CI_SLAB_INIT": 1, "CI_SLAB_STORE": 1, "CI_STAGE_DEEP": 0}
DROP_DEFAULT_DEFINE = set(VALUE)      # "#ifndef X / #define X v / #endif" blocks of the VALUE macros disappear

def decide(line):
    m = re.match(r"#\s*ifdef\s+(\w+)", line)
    if m: return False if m.group(1) in UNDEF else (bool(VALUE[m.group(1)]) or True if m.group(1) in VALUE else None)
    m = re.match(r"#\s*ifndef\s+(\w+)", line)
    if m: return True if m.group(1) in UNDEF else (False if m.group(1) in VALUE else None)
    m = re.match(r"#\s*if\s+(.*?)\s*(//.*)?$", line)
    if m:
        e = m.group(1)
        names = set(re.findall(r"defined\((\w+)\)", e)) | set(re.findall(r"\b([A-Z][A-Z0-9_]+)\b", e))
        names -= {"defined"}
        if names and names <= (UNDEF | set(VALUE)):
            py = re.sub(r"defined\((\w+)\)", lambda mm: "True" if mm.group(1) in VALUE else "False", e)
            for k, v in VALUE.items(): py = re.sub(r"\b%s\b" % k, str(v), py)
            py = py.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
            return bool(eval(py))
        return None
    return None

src = open(sys.argv[1]).read().split("\n")
out, stack = [], []      # stack entries: (decision or None, currently_emitting_parent, taken_branch_seen)
emit = True
i = 0
while i < len(src):
    ln = src[i]
    st = ln.lstrip()
    if re.match(r"#\s*(if|ifdef|ifndef)\b", st):
        d = decide(st)
        if d is False and re.match(r"#\s*ifndef\s+(\w+)", st) and re.match(r"#\s*ifndef\s+(\w+)", st).group(1) in VALUE:
            pass
        stack.append([d, emit, False])
        if d is None:
            if emit: out.append(ln)
        else:
            emit = emit and d
            stack[-1][2] = d
        i += 1; continue
    if re.match(r"#\s*else\b", st) and stack:
        d, parent, taken = stack[-1]
        if d is None:
            if emit: out.append(ln)
        else:
            emit = parent and not taken
        i += 1; continue
    if re.match(r"#\s*elif\b", st) and stack and stack[-1][0] is not None:
        raise SystemExit("elif on a resolved conditional: not handled")
    if re.match(r"#\s*endif\b", st) and stack:
        d, parent, taken = stack.pop()
        if d is None:
            if emit: out.append(ln)
        emit = parent
        i += 1; continue
    if emit: out.append(ln)
    i += 1
open(sys.argv[2], "w").write("\n".join(out))
