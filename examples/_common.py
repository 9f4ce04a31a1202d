"""Shared pieces of the two examples: the reference's prompts (examples/audio_input_beat_detection.rs:60-112)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import beat_detector_amd as bd  # noqa: E402
from beat_detector_amd import record  # noqa: E402


def read_number(stdin) -> int:
    line = ""
    while not line:
        line = stdin.readline()
        if line == "":
            raise SystemExit("Input must be a valid number!")
    try:
        value = int(line.strip())
    except ValueError:
        raise SystemExit("Input must be a valid number!")
    if not 0 <= value <= 255:           # the reference parses a u8
        raise SystemExit("Input must be a valid number!")
    return value


def select_input_device(devs, stdin=sys.stdin, out=sys.stdout):
    print("Available audio devices:", file=out)
    for i, name in enumerate(devs):
        print(f"  [{i}] {name}", file=out)
    print("Select audio device: input device number and enter:", file=out)
    choice = read_number(stdin)
    return list(devs.values())[choice]      # IndexError = the reference's unwrap() on None


def select_strategy(stdin=sys.stdin, out=sys.stdout):
    print("Available beat detection strategies:", file=out)
    for i, s in enumerate(bd.StrategyKind.values()):
        print(f"  [{i}] {s.name_()} - {s.description()}", file=out)
    print("Select strategy: input id and enter:", file=out)
    choice = read_number(stdin)
    if choice == 0:
        return bd.StrategyKind.LPF
    if choice == 1:
        return bd.StrategyKind.Spectrum
    raise SystemExit("Invalid strategy!")


def listen(dev, strategy, out=sys.stdout):
    """The body both examples share: Ctrl-C handler, on_beat printing, start_listening, join."""
    import ctypes
    import signal
    recording = ctypes.c_int(1)

    def stop(signum, frame):
        print("Stopping recording", file=sys.stderr)
        recording.value = 0

    try:
        signal.signal(signal.SIGINT, stop)
    except ValueError:                      # not the main thread (tests)
        pass
    handle = bd.start_listening(lambda info: print(f"Found beat at {info!r}ms", file=out, flush=True), dev, strategy, recording)
    handle.join()
