"""Generator of tests/golden/trainer_rela_calls.json: every call the reference's UNMODIFIED trainer makes into `cfvpy.rela`
(cfvpy/selfplay.py, cfvpy/utils.py), as data -- callee, number of positional arguments, keyword names, source line -- taken from
the files' syntax trees.  Run in the build container (it reads /root/reference); the fixture travels, the reference does not.
usage: python tests/golden/make_trainer_calls.py"""
import ast
import json
import os

REF = "/root/reference/cfvpy"
# receivers the trainer binds to rela objects (selfplay.py:219-252: model_locker / replay / policy_replay / context)
RECEIVERS = {"replay": "ValuePrioritizedReplay", "policy_replay": "ValuePrioritizedReplay", "context": "Context",
             "model_locker": "ModelLocker"}
METHODS = {"ValuePrioritizedReplay": {"size", "num_add", "sample", "pop_until", "load", "save", "extract", "push", "update_priority"},
           "Context": {"push_env_thread", "start", "pause", "resume", "terminate", "terminated"},
           "ModelLocker": {"update_model"}}


def dotted(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
    return ".".join(reversed(parts))


calls, bases = [], []
for fname in ("selfplay.py", "utils.py"):
    tree = ast.parse(open(os.path.join(REF, fname)).read())
    # dictionaries built with dict(k=...) and later splatted into a call (selfplay.py:222-229 `replay_params`)
    dicts = {t.id: [k.arg for k in node.value.keywords if k.arg] for node in ast.walk(tree) if isinstance(node, ast.Assign)
             and isinstance(node.value, ast.Call) and dotted(node.value.func) == "dict" for t in node.targets if isinstance(t, ast.Name)}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef):
            for b in node.bases:
                if dotted(b).endswith("rela.Context"):
                    bases.append({"file": fname, "line": node.lineno, "class": node.name, "base": "Context"})
        if not isinstance(node, ast.Call):
            continue
        name = dotted(node.func)
        callee = None
        if ".rela." in name or name.startswith("rela."):
            callee = name.split("rela.", 1)[1]
        else:
            recv, _, meth = name.rpartition(".")
            cls = RECEIVERS.get(recv.split(".")[-1])
            if cls and meth in METHODS[cls]:
                callee = f"{cls}.{meth}"
        if callee is None:
            continue
        calls.append({"file": fname, "line": node.lineno, "callee": callee, "n_positional": len(node.args),
                      "keywords": [k.arg for k in node.keywords if k.arg], "star_kwargs_keys": [key for k in node.keywords if k.arg is None and isinstance(k.value, ast.Name)
                                            for key in dicts.get(k.value.id, [])]})
out = {"source": "cfvpy/selfplay.py + cfvpy/utils.py of /root/reference (facebookresearch/rebel), syntax trees only",
       "calls": sorted(calls, key=lambda c: (c["file"], c["line"])), "subclasses": bases}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trainer_rela_calls.json")
json.dump(out, open(path, "w"), indent=1)
print(len(calls), "calls,", len(bases), "subclasses ->", path)
