"""`ops.precision(...)` is a per-thread (context-variable) scope, like the thread-local scope of the library's own entry points
(include/rlt_hip.h): a scope entered on one thread is invisible on another, nested scopes restore the outer one, and one scope
object can be re-entered.  No GPU needed: the scope is host state, the process default is read from the loaded library."""
import os
import sys
import threading

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ranked-list-truncation_amd"))


def test_precision_scope_is_per_thread_and_reentrant():
    from rlt_hip import native as N
    from rlt_hip import ops
    default = int(N.load().rlt_get_precision())
    assert ops.current_precision() == default
    entered, release = threading.Event(), threading.Event()
    seen = {}

    def worker():
        with ops.precision("fp32"):
            seen["inside"] = ops.current_precision()
            entered.set()
            release.wait(10)
            with ops.precision("bf16x3"):
                seen["nested"] = ops.current_precision()
            seen["restored"] = ops.current_precision()
        seen["after"] = ops.current_precision()

    t = threading.Thread(target=worker)
    t.start()
    assert entered.wait(10)
    # the worker sits inside its fp32 scope: this thread still sees the process default, and its own scope does not leak over
    assert ops.current_precision() == default
    scope = ops.precision("bf16x6")
    with scope:
        assert ops.current_precision() == N.PRECISION_BF16X6
        with scope:                                  # the same object again: a stack of tokens, not one saved value
            assert ops.current_precision() == N.PRECISION_BF16X6
        assert ops.current_precision() == N.PRECISION_BF16X6
        release.set()
        t.join(10)
        assert ops.current_precision() == N.PRECISION_BF16X6
    assert ops.current_precision() == default
    assert seen == {"inside": N.PRECISION_FP32, "nested": N.PRECISION_BF16X3, "restored": N.PRECISION_FP32, "after": default}
