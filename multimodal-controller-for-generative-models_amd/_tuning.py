"""A/B switches of the tuning runs (tools/ab_bench.sh and friends).

The shipped path has NO environment-dependent dispatch: every `MCGEN_*` switch below is honoured only when the process
was started with ``MCGEN_TUNING=1``; otherwise each switch takes its default whatever the environment holds, so a
variable left over in a driver's environment cannot change what bench.py times.  `ACTIVE` lists the switches that
were read with a non-default value (bench.py prints it as `tuning_switches`).
"""
import os

ENABLED = os.environ.get('MCGEN_TUNING', '0') == '1'
ACTIVE = {}


def flag(name: str, default: str) -> str:
    """The value of switch `name` ('MCGEN_...'): the environment's under MCGEN_TUNING=1, else `default`."""
    if not ENABLED:
        return default
    v = os.environ.get(name, default)
    if v != default:
        ACTIVE[name] = v
    return v
