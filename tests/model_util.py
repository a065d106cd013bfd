"""The detecting-head checkpoint lives with the oracle (oracle/calibrate.py: it calibrates the head WITH the oracle forward, and
__graft_entry__.smoke() must not depend on tests/); kept importable under its old name for the tests."""
from oracle.calibrate import detecting_state_dict  # noqa: F401
