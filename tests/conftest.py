import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bounded-lsq_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


import pytest  # noqa: E402


@pytest.fixture
def blsq_opt(monkeypatch):
    """blsq_opt(name, value, ctx=None): one switch of the library's option table for the rest of the test.
    `name`: the table's key or its environment variable.  The library reads its switches ONCE per ctx, at
    blsq_ctx_create; so the fixture (i) sets the environment variable — every ctx the test creates afterwards starts
    with it — and (ii) sets the option on `ctx` and on the shared default contexts that exist already.  When the test
    ends every context touched — and every default context created meanwhile — gets the value back that the
    environment of the test session prescribes."""
    touched = {}                                               # option -> (value the session's environment gives it)
    ctxs = []

    def setter(name, value, ctx=None):
        from bounded_lsq import _hip_step
        envname = name if name.startswith("BLSQ_") else "BLSQ_" + name.upper()
        if name not in touched:
            touched[name] = os.environ.get(envname)
        monkeypatch.setenv(envname, value if isinstance(value, str) else repr(float(value)))
        for c in ([ctx] if ctx is not None else []) + list(_hip_step._default_ctx.values()):
            c.set_option(name, float(value))
            if c not in ctxs:
                ctxs.append(c)
    yield setter
    from bounded_lsq import _hip_step
    for c in ctxs + [c for c in _hip_step._default_ctx.values() if c not in ctxs]:
        try:
            table = {o["name"]: o["default"] for o in c.options()}
            table.update({o["env"]: o["default"] for o in c.options()})
            for name, env0 in touched.items():
                c.set_option(name, float(env0) if env0 not in (None, "") else table[name])
        except Exception:                                      # noqa: BLE001  (a ctx the test has closed)
            pass
