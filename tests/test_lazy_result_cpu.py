"""CPU-only: pipeline._LazyResult -- render()'s result dict with entries computed on first access (the reference's
`visibility_filter_d/_s`, gaussian_renderer/__init__.py:199-203, cost a host sync each and are never used by its callers).
A caller that copies or merges the dict must still get the values, never the internal placeholder."""


def _make():
    from s3gaussian_amd.pipeline import _LazyResult
    calls = []
    out = _LazyResult({"render": 1})
    out.set_lazy("visibility_filter_d", lambda: calls.append("d") or "D")
    out.set_lazy("visibility_filter_s", lambda: calls.append("s") or "S")
    return out, calls


def test_lazy_entries_are_not_computed_until_read():
    out, calls = _make()
    assert out["render"] == 1 and calls == [] and "visibility_filter_d" in out and len(out) == 3
    assert out["visibility_filter_d"] == "D" and calls == ["d"]
    assert out.get("visibility_filter_s") == "S" and out["visibility_filter_d"] == "D" and calls == ["d", "s"]


def test_copies_and_merges_see_the_values_not_the_placeholders():
    want = {"render": 1, "visibility_filter_d": "D", "visibility_filter_s": "S"}
    for how in (lambda o: dict(o), lambda o: {**o}, lambda o: o.copy(), lambda o: dict(o.items()), lambda o: o | {},
                lambda o: {} | o, lambda o: {k: o[k] for k in o}, lambda o: dict(zip(o.keys(), o.values()))):
        out, _ = _make()
        assert how(out) == want
    out, _ = _make()
    assert out.pop("visibility_filter_d") == "D" and out.setdefault("visibility_filter_s", 7) == "S"
    out, calls = _make()
    out["visibility_filter_d"] = "override"          # an explicit store wins and drops the thunk
    assert dict(out)["visibility_filter_d"] == "override" and calls == ["s"]
