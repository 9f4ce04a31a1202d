//! Raw declarations of `include/beat_detector_hip.h`, one per entry point, in the
//! header's order.  Types keep their C names so that a declaration can be compared
//! with the header line by line (`tests/test_cabi_symbols.py` does exactly that).
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const BD_OK: c_int = 0;
pub const BD_ERR_INVALID_ARG: c_int = -1;
pub const BD_ERR_NO_DEVICE: c_int = -2;
pub const BD_ERR_HIP: c_int = -3;
pub const BD_ERR_UNSUPPORTED: c_int = -4;
pub const BD_ERR_EMPTY_WINDOW: c_int = -5;
pub const BD_ERR_NO_BINS: c_int = -6;
pub const BD_ERR_STOPPED: c_int = -7;

pub const BD_KIND_LPF: c_int = 0;
pub const BD_KIND_SPECTRUM: c_int = 1;

pub const BD_DIV_BINS: u32 = 0;
pub const BD_DIV_FFT_LEN: u32 = 1;
pub const BD_DIV_NONE: u32 = 2;

pub const BD_FMT_I16: c_int = 0;
pub const BD_FMT_U16: c_int = 1;
pub const BD_FMT_F32: c_int = 2;

pub const BD_NOALLOC_SCRATCH_BYTES: u64 = 96 << 20;

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct bd_params {
    pub min_amplitude: u32,
    pub min_beat_gap_ms: u32,
    pub spectrum_max_hz: f32,
    pub spectrum_threshold: f32,
    pub spectrum_divisor: u32,
    pub spectrum_include_dc: u32,
    pub spectrum_inclusive: u32,
    pub spectrum_dc_packs_nyquist: u32,
    pub lpf_cutoff_hz: u32,
    pub lpf_ratio: f32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct bd_onset_params {
    pub fft_len: u32,
    pub hop: u32,
    pub window: u32,
    pub band_max_hz: f32,
    pub mean_frames: u32,
    pub max_frames: u32,
    pub delta: f32,
    pub lambda: f32,
    pub min_gap_ms: u32,
    pub use_median: u32,
    pub flux_power: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct bd_biquad_design {
    pub b0: f32,
    pub b1: f32,
    pub b2: f32,
    pub na1: f32,
    pub na2: f32,
    pub c1b: f32,
    pub c2b: f32,
    pub pw: [[f32; 4]; 6],
    pub q: [[f32; 4]; 64],
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct bd_biquad_params {
    pub cutoff_hz: u32,
    pub q_factor: f32,
    pub window_len: u32,
    pub ratio: f32,
    pub min_amplitude: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct bd_synth_source {
    pub seed: u64,
    pub first_stream: u64,
    pub sampling_rate: u32,
}

pub type bd_read_fn = unsafe extern "C" fn(user: *mut c_void, buffer: *mut c_void, capacity: usize) -> usize;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct bd_input_source {
    pub sampling_rate: u32,
    pub sample_format: c_int,
    pub max_callback_len: u32,
    pub read: Option<bd_read_fn>,
    pub user: *mut c_void,
}

// opaque handles
#[repr(C)]
pub struct bd_detector {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bd_live {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bd_listener {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bd_chunked {
    _private: [u8; 0],
}

pub type bd_live_on_beat_fn = Option<unsafe extern "C" fn(user: *mut c_void, stream: u32, relative_ms: u32)>;
pub type bd_on_beat_fn = Option<unsafe extern "C" fn(user: *mut c_void, relative_ms: u32)>;
pub type bd_chunk_source_fn = Option<
    unsafe extern "C" fn(
        user: *mut c_void,
        dst: *mut i16,
        first_stream: u64,
        n_streams: u64,
        n_samples: u64,
        stream_stride: u64,
        hip_stream: *mut c_void,
    ) -> c_int,
>;

extern "C" {
    pub fn bd_default_params(p: *mut bd_params);
    pub fn bd_strerror(status: c_int) -> *const c_char;
    pub fn bd_last_error() -> *const c_char;
    pub fn bd_device_count(count: *mut c_int) -> c_int;
    pub fn bd_current_device(device: *mut c_int) -> c_int;
    pub fn bd_set_device(device: c_int) -> c_int;
    pub fn bd_device_name(device: c_int, name: *mut c_char, capacity: usize, total_memory_bytes: *mut u64) -> c_int;

    // StrategyKind metadata
    pub fn bd_kind_count() -> c_int;
    pub fn bd_kind_name(kind: c_int) -> *const c_char;
    pub fn bd_kind_description(kind: c_int) -> *const c_char;
    pub fn bd_kind_min_duration_between_beats_ms(kind: c_int) -> u32;

    // per-window detector
    pub fn bd_detector_create(kind: c_int, sampling_rate: u32, params: *const bd_params, device: c_int, out: *mut *mut bd_detector) -> c_int;
    pub fn bd_detector_is_beat(d: *mut bd_detector, samples: *const i16, n: usize, relative_ms: *mut u32) -> c_int;
    pub fn bd_detector_kind(d: *const bd_detector) -> c_int;
    pub fn bd_detector_last(d: *const bd_detector, value: *mut f32, candidate: *mut c_int) -> c_int;
    pub fn bd_detector_destroy(d: *mut bd_detector);

    // batched offline analysis
    pub fn bd_batch_callbacks(n_samples: u64, callback_len: u32) -> u64;
    pub fn bd_batch_max_beats(kind: u32, sampling_rate: u32, params: *const bd_params, n_samples: u64, callback_len: u32) -> u32;
    pub fn bd_analyze_batch(kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *const i16, n_streams: u64, n_samples: u64, stream_stride: u64, callback_len: u32, values: *mut f32, flags: *mut u8, out_stride: u64, beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, hip_stream: *mut c_void) -> c_int;
    pub fn bd_analyze_batch_ragged(kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *const i16, n_streams: u64, lengths: *const u64, max_samples: u64, stream_stride: u64, callback_len: u32, values: *mut f32, flags: *mut u8, out_stride: u64, beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, hip_stream: *mut c_void) -> c_int;
    pub fn bd_batch_candidates(kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *const i16, n_streams: u64, n_samples: u64, stream_stride: u64, callback_len: u32, values: *mut f32, flags: *mut u8, out_stride: u64, hip_stream: *mut c_void) -> c_int;
    pub fn bd_analyze_batch_host(kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *const i16, n_streams: u64, n_samples: u64, stream_stride: u64, callback_len: u32, values: *mut f32, flags: *mut u8, beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, device: c_int) -> c_int;
    pub fn bd_analyze_batch_multi(kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *const i16, n_streams: u64, n_samples: u64, stream_stride: u64, callback_len: u32, values: *mut f32, flags: *mut u8, beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, devices: *const c_int, n_devices: c_int) -> c_int;
    pub fn bd_analyze_batch_devices(kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *mut *const i16, n_streams: *const u64, n_samples: u64, stream_stride: u64, callback_len: u32, values: *mut *mut f32, flags: *mut *mut u8, out_stride: u64, beat_count: *mut *mut u32, beat_idx: *mut *mut u32, beat_ms: *mut *mut u32, max_beats: u32, devices: *const c_int, n_devices: c_int) -> c_int;
    pub fn bd_refractory(sampling_rate: u32, params: *const bd_params, callback_len: u32, flags: *const u8, flags_stride: u64, n_streams: u64, n_callbacks: u64, beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, hip_stream: *mut c_void) -> c_int;
    pub fn bd_analyze_batch_noalloc(kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *const i16, n_streams: u64, n_samples: u64, stream_stride: u64, callback_len: u32, beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, hip_stream: *mut c_void) -> c_int;

    // live path
    pub fn bd_live_create(kind: c_int, sampling_rate: u32, params: *const bd_params, n_streams: u32, max_callback_len: u32, depth: u32, device: c_int, out: *mut *mut bd_live) -> c_int;
    pub fn bd_live_set_on_beat(live: *mut bd_live, on_beat: bd_live_on_beat_fn, user: *mut c_void) -> c_int;
    pub fn bd_live_submit(live: *mut bd_live, samples: *const c_void, n: u32, format: c_int, tick_id: *mut u64) -> c_int;
    pub fn bd_live_submit_ragged(live: *mut bd_live, samples: *const c_void, n_per_stream: *const u32, row_pitch: u32, format: c_int, tick_id: *mut u64) -> c_int;
    pub fn bd_live_poll(live: *mut bd_live, tick_id: u64, wait: c_int, beats: *mut u8, relative_ms: *mut u32, values: *mut f32) -> c_int;
    pub fn bd_live_destroy(live: *mut bd_live);
    pub fn bd_start_listening(on_beat: bd_on_beat_fn, on_beat_user: *mut c_void, source: *const bd_input_source, kind: c_int, keep_recording: *const c_int, out: *mut *mut bd_listener) -> c_int;
    pub fn bd_listener_join(listener: *mut bd_listener) -> c_int;

    // ingest converters (device pointers)
    pub fn bd_u16_data_to_i16(input: *const u16, out: *mut i16, n: u64, hip_stream: *mut c_void) -> c_int;
    pub fn bd_f32_data_to_i16(input: *const f32, out: *mut i16, n: u64, hip_stream: *mut c_void) -> c_int;
    pub fn bd_stereo_to_mono(interleaved: *const i16, mono: *mut i16, n_frames: u64, hip_stream: *mut c_void) -> c_int;

    // synthetic click tracks
    pub fn bd_synth_fill(samples: *mut i16, n_streams: u64, n_samples: u64, stream_stride: u64, first_stream: u64, seed: u64, sampling_rate: u32, hip_stream: *mut c_void) -> c_int;
    pub fn bd_synth_fill_host(samples: *mut i16, n_streams: u64, n_samples: u64, stream_stride: u64, first_stream: u64, seed: u64, sampling_rate: u32) -> c_int;

    // extension analysis: STFT onset detection (no counterpart in the parent crate)
    pub fn bd_onset_default_params(p: *mut bd_onset_params);
    pub fn bd_onset_frames(n_samples: u64, fft_len: u32, hop: u32) -> u64;
    pub fn bd_onset_stft(sampling_rate: u32, p: *const bd_onset_params, samples: *const i16, n_streams: u64, n_samples: u64, stream_stride: u64, flux: *mut f32, band_energy: *mut f32, out_stride: u64, hip_stream: *mut c_void) -> c_int;
    pub fn bd_onset_pick(sampling_rate: u32, p: *const bd_onset_params, flux: *const f32, n_streams: u64, n_frames: u64, flux_stride: u64, candidates: *mut u8, onset_count: *mut u32, onset_frame: *mut u32, onset_ms: *mut u32, max_onsets: u32, hip_stream: *mut c_void) -> c_int;
    pub fn bd_onset_batch(sampling_rate: u32, p: *const bd_onset_params, samples: *const i16, n_streams: u64, n_samples: u64, stream_stride: u64, flux: *mut f32, band_energy: *mut f32, out_stride: u64, onset_count: *mut u32, onset_frame: *mut u32, onset_ms: *mut u32, max_onsets: u32, hip_stream: *mut c_void) -> c_int;

    // extension analysis: carried-state biquad envelope (no counterpart in the parent crate)
    pub fn bd_biquad_default_params(p: *mut bd_biquad_params);
    pub fn bd_biquad_design_lowpass(sampling_rate: u32, cutoff_hz: u32, q_factor: f32, d: *mut bd_biquad_design) -> c_int;
    pub fn bd_biquad_envelope(sampling_rate: u32, params: *const bd_biquad_params, design: *const bd_biquad_design, samples: *const i16, n_streams: u64, n_samples: u64, stream_stride: u64, values: *mut f32, flags: *mut u8, out_stride: u64, hip_stream: *mut c_void) -> c_int;

    // chunked executor: shards larger than device memory
    pub fn bd_chunk_source_synth(user: *mut c_void, dst: *mut i16, first_stream: u64, n_streams: u64, n_samples: u64, stream_stride: u64, hip_stream: *mut c_void) -> c_int;
    pub fn bd_chunked_create(chunk_streams: u64, n_samples: u64, device: c_int, out: *mut *mut bd_chunked) -> c_int;
    pub fn bd_chunked_device_bytes(ex: *const bd_chunked) -> u64;
    pub fn bd_chunked_analyze(ex: *mut bd_chunked, kind: c_int, sampling_rate: u32, params: *const bd_params, callback_len: u32, source: bd_chunk_source_fn, user: *mut c_void, n_streams: u64, beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, hip_stream: *mut c_void) -> c_int;
    pub fn bd_chunked_onset(ex: *mut bd_chunked, sampling_rate: u32, p: *const bd_onset_params, source: bd_chunk_source_fn, user: *mut c_void, n_streams: u64, onset_count: *mut u32, onset_frame: *mut u32, onset_ms: *mut u32, max_onsets: u32, hip_stream: *mut c_void) -> c_int;
    pub fn bd_chunked_last_timing(ex: *mut bd_chunked, main_ms: *mut f32, post_ms: *mut f32, capacity: u32) -> c_int;
    pub fn bd_chunked_destroy(ex: *mut bd_chunked);

    // instrumentation
    pub fn bd_last_kernel_name() -> *const c_char;
    pub fn bd_tuning_reload();
    pub fn bd_probe_read(src: *const c_void, bytes: u64, sink: *mut u32, waves_per_cu: u32, hip_stream: *mut c_void) -> c_int;
    pub fn bd_probe_read_ex(src: *const c_void, bytes: u64, sink: *mut u32, waves_per_cu: u32, depth: u32, contiguous: u32, hip_stream: *mut c_void) -> c_int;
}
