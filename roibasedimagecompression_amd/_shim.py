"""Helpers of the `encoder/` and `decoder/` import-surface packages.

The reference's `encoder` / `decoder` are implicit namespace packages; this repository ships regular packages of the
same names that hold the MI355X modules of the hot path.  Two mechanisms keep every OTHER module of the reference
reachable when both trees are on `sys.path` (this repository first):

* every shim package extends its `__path__` over the same-named directories further along `sys.path`
  (`pkgutil.extend_path`), so `encoder.ROI.thin_regions2`, `other.*` ... resolve to the reference's files;
* a placeholder module (a stage this build does not implement) calls `defer_to_downstream()` first: if a real module of
  the same name exists in a later portion of its package, that file is executed in its place.
"""
import importlib.util
import os
import sys


def defer_to_downstream(module_name, module_file):
    """Execute the same-named module found further along the parent package's __path__ (i.e. the reference's own file)
    in place of the calling placeholder.  Returns the module, or None when there is none (the placeholder stays)."""
    pkg_name, _, leaf = module_name.rpartition(".")
    pkg = sys.modules.get(pkg_name)
    if pkg is None or not hasattr(pkg, "__path__"):
        return None
    here = os.path.dirname(os.path.abspath(module_file))
    for d in list(pkg.__path__):
        if os.path.abspath(d) == here:
            continue
        cand = os.path.join(d, leaf + ".py")
        if os.path.isfile(cand):
            spec = importlib.util.spec_from_file_location(module_name, cand)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[module_name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(module_name, None)
                raise
            return mod
    return None


def upstream(name):
    """a function of a stage outside this build: raises with a pointer to INTEGRATION.md"""
    def fn(*args, **kwargs):
        raise NotImplementedError(f"{name}: stage outside the MI355X hot path -- put the reference checkout behind this "
                                  "repository on sys.path and its own module is used (see INTEGRATION.md)")
    fn.__name__ = name
    return fn


def downstream_getattr(module_name, module_file, helpers):
    """Module-level `__getattr__` of a mirror that implements PART of a reference module: the names in `helpers` come from
    the reference's own file when it sits further along the package's __path__ (and its imports resolve), otherwise they
    are placeholders that raise when called."""
    state = {}

    def load():
        if "mod" in state:
            return state["mod"]
        state["mod"] = None
        pkg = sys.modules.get(module_name.rpartition(".")[0])
        here = os.path.dirname(os.path.abspath(module_file))
        leaf = module_name.rpartition(".")[2]
        for d in list(getattr(pkg, "__path__", [])):
            cand = os.path.join(d, leaf + ".py")
            if os.path.abspath(d) != here and os.path.isfile(cand):
                spec = importlib.util.spec_from_file_location(module_name + "._reference", cand)
                mod = importlib.util.module_from_spec(spec)
                try:
                    spec.loader.exec_module(mod)
                except ImportError:                              # the reference's file needs OpenCV / scikit-image
                    mod = None
                state["mod"] = mod
                break
        return state["mod"]

    def getattr_(name):
        if name in helpers:
            mod = load()
            fn = getattr(mod, name) if mod is not None and hasattr(mod, name) else upstream(name)
            sys.modules[module_name].__dict__[name] = fn
            return fn
        raise AttributeError(name)
    return getattr_
