"""Counterpart of the reference's examples/minimal.rs: first input device, Spectrum strategy,
prints every beat.  `python examples/minimal.py` (needs a GPU)."""
from _common import bd, listen, record


def select_input_device():
    return next(iter(record.audio_input_device_list(realtime=True).values()))


def select_strategy():
    return bd.StrategyKind.Spectrum


if __name__ == "__main__":
    listen(select_input_device(), select_strategy())
