"""Counterpart of the reference's examples/audio_input_beat_detection.rs: choose an input device
and a strategy on the console, print every beat until the input ends or Ctrl-C.

    BEATDET_INPUTS=/path/to/wavs python examples/audio_input_beat_detection.py
    arecord -f S16_LE -r 44100 -c 1 -t raw | python examples/audio_input_beat_detection.py --stdin-pcm 1
"""
import sys

from _common import bd, listen, record, select_input_device, select_strategy  # noqa: F401


def main(argv=None, stdin=sys.stdin, out=sys.stdout):
    argv = sys.argv[1:] if argv is None else argv
    if argv[:1] == ["--stdin-pcm"]:
        # samples arrive on stdin, so the strategy comes from the command line
        strategy = bd.StrategyKind.values()[int(argv[1])]
        dev = record.RawPcmDevice(sys.stdin.buffer, name="stdin")
    else:
        devs = record.audio_input_device_list(realtime="--fast" not in argv)
        if not devs:
            raise SystemExit("No audio input devices found!")
        dev = select_input_device(devs, stdin, out) if len(devs) > 1 else next(iter(devs.values()))
        strategy = select_strategy(stdin, out)
    listen(dev, strategy, out)


if __name__ == "__main__":
    main()
