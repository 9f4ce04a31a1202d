"""MI355X-native implementation of beat-detector's per-window analysis path.

Python mirror of the reference's public interface for this path
(`StrategyKind`, `Strategy::is_beat`, `BeatInfo`; src/lib.rs:51-185 of the
reference) over the C ABI declared in ``include/beat_detector_hip.h``, plus the
batched entry points.  All computation happens in ``libbeatdet_hip.so`` on the
GPU; there is no CPU implementation in this package and every compute call
fails loudly when the library or a GPU is missing.
"""
from .api import (  # noqa: F401
    BeatInfo,
    BiquadDesign,
    BiquadParams,
    BeatDetectorError,
    FMT_F32,
    FMT_I16,
    FMT_U16,
    Live,
    OnsetParams,
    Params,
    Strategy,
    StrategyKind,
    analyze_batch,
    analyze_batch_host,
    analyze_batch_multi,
    analyze_batch_devices,
    analyze_batch_noalloc,
    NOALLOC_MAX_CALLBACKS,
    analyze_batch_ragged,
    batch_callbacks,
    batch_candidates,
    batch_max_beats,
    device_count,
    device_name,
    f32_data_to_i16,
    last_kernel_name,
    lib,
    lib_path,
    refractory,
    start_listening,
    stereo_to_mono,
    synth_fill,
    synth_fill_host,
    tuning,
    u16_data_to_i16,
)
from .onset import OnsetResult, onset_batch, onset_frames, onset_pick, onset_stft  # noqa: F401,E402
from .biquad import biquad_beats, biquad_design, biquad_envelope  # noqa: F401,E402
from .executor import ChunkedExecutor, SynthSource  # noqa: F401,E402
from . import record  # noqa: F401,E402
