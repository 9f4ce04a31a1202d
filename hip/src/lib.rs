//! Raw bindings and thin safe wrappers for `libbeatdet_hip.so`
//! (`include/beat_detector_hip.h`): the GPU implementation of beat-detector's
//! per-window analysis path.  SOURCE ONLY, not compiled in this repository's
//! build environment.
//!
//! The parent crate keeps its `Strategy` trait, `StrategyKind` enum and
//! `record::start_listening` signature; `StrategyKind::detector` returns a
//! type that wraps [`Detector`].
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const BD_OK: c_int = 0;
pub const BD_KIND_LPF: c_int = 0;
pub const BD_KIND_SPECTRUM: c_int = 1;

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
pub struct bd_detector {
    _private: [u8; 0],
}

extern "C" {
    pub fn bd_default_params(p: *mut bd_params);
    pub fn bd_last_error() -> *const c_char;
    pub fn bd_strerror(status: c_int) -> *const c_char;
    pub fn bd_device_count(count: *mut c_int) -> c_int;
    pub fn bd_kind_count() -> c_int;
    pub fn bd_kind_name(kind: c_int) -> *const c_char;
    pub fn bd_kind_description(kind: c_int) -> *const c_char;
    pub fn bd_kind_min_duration_between_beats_ms(kind: c_int) -> u32;
    pub fn bd_detector_create(
        kind: c_int,
        sampling_rate: u32,
        params: *const bd_params,
        device: c_int,
        out: *mut *mut bd_detector,
    ) -> c_int;
    pub fn bd_detector_is_beat(d: *mut bd_detector, samples: *const i16, n: usize, relative_ms: *mut u32) -> c_int;
    pub fn bd_detector_kind(d: *const bd_detector) -> c_int;
    pub fn bd_detector_destroy(d: *mut bd_detector);
    pub fn bd_batch_callbacks(n_samples: u64, callback_len: u32) -> u64;
    pub fn bd_batch_max_beats(kind: u32, sampling_rate: u32, params: *const bd_params, n_samples: u64, callback_len: u32) -> u32;
    pub fn bd_analyze_batch(
        kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *const i16, n_streams: u64,
        n_samples: u64, stream_stride: u64, callback_len: u32, values: *mut f32, flags: *mut u8, out_stride: u64,
        beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, hip_stream: *mut c_void,
    ) -> c_int;
    pub fn bd_analyze_batch_host(
        kind: c_int, sampling_rate: u32, params: *const bd_params, samples: *const i16, n_streams: u64,
        n_samples: u64, stream_stride: u64, callback_len: u32, values: *mut f32, flags: *mut u8,
        beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32, max_beats: u32, device: c_int,
    ) -> c_int;
    pub fn bd_refractory(
        sampling_rate: u32, params: *const bd_params, callback_len: u32, flags: *const u8, flags_stride: u64,
        n_streams: u64, n_callbacks: u64, beat_count: *mut u32, beat_idx: *mut u32, beat_ms: *mut u32,
        max_beats: u32, hip_stream: *mut c_void,
    ) -> c_int;
    // Extension analyses: no counterpart in the parent crate (device pointers, enqueue only).
    pub fn bd_onset_default_params(p: *mut bd_onset_params);
    pub fn bd_onset_frames(n_samples: u64, fft_len: u32, hop: u32) -> u64;
    pub fn bd_onset_batch(
        sampling_rate: u32, p: *const bd_onset_params, samples: *const i16, n_streams: u64, n_samples: u64,
        stream_stride: u64, flux: *mut f32, band_energy: *mut f32, out_stride: u64, onset_count: *mut u32,
        onset_frame: *mut u32, onset_ms: *mut u32, max_onsets: u32, hip_stream: *mut c_void,
    ) -> c_int;
    pub fn bd_biquad_default_params(p: *mut bd_biquad_params);
    pub fn bd_biquad_envelope(
        sampling_rate: u32, params: *const bd_biquad_params, design: *const c_void, samples: *const i16,
        n_streams: u64, n_samples: u64, stream_stride: u64, values: *mut f32, flags: *mut u8, out_stride: u64,
        hip_stream: *mut c_void,
    ) -> c_int;
}

/// `bd_onset_params` of the C header.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
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
}

/// `bd_biquad_params` of the C header.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bd_biquad_params {
    pub cutoff_hz: u32,
    pub q_factor: f32,
    pub window_len: u32,
    pub ratio: f32,
    pub min_amplitude: u32,
}

/// Message of the last failed call on this thread.
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(bd_last_error()).to_string_lossy().into_owned() }
}

/// Owning handle of one GPU detector (`StrategyKind::detector`).
#[derive(Debug)]
pub struct Detector {
    raw: *mut bd_detector,
}

// The reference's detectors are `Send` but not `Sync` (src/lib.rs:155); same here.
unsafe impl Send for Detector {}

impl Detector {
    pub fn new(kind: c_int, sampling_rate: u32) -> Result<Self, String> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { bd_detector_create(kind, sampling_rate, std::ptr::null(), -1, &mut raw) };
        if rc == BD_OK {
            Ok(Self { raw })
        } else {
            Err(last_error())
        }
    }

    /// `Strategy::is_beat`: `Some(relative_ms)` on a beat.
    pub fn is_beat(&self, samples: &[i16]) -> Result<Option<u32>, String> {
        let mut ms = 0u32;
        match unsafe { bd_detector_is_beat(self.raw, samples.as_ptr(), samples.len(), &mut ms) } {
            1 => Ok(Some(ms)),
            0 => Ok(None),
            _ => Err(last_error()),
        }
    }
}

impl Drop for Detector {
    fn drop(&mut self) {
        unsafe { bd_detector_destroy(self.raw) }
    }
}

/// Beats of many equally long streams (stream-major `samples`, `n_samples` each)
/// analysed in one batch on the current GPU: per stream, `(callback index, relative_ms)`.
pub fn analyze_batch(kind: c_int, sampling_rate: u32, samples: &[i16], n_samples: usize, callback_len: u32)
    -> Result<Vec<Vec<(u32, u32)>>, String> {
    assert!(n_samples > 0 && samples.len() % n_samples == 0);
    let n_streams = samples.len() / n_samples;
    let max_beats = unsafe { bd_batch_max_beats(kind as u32, sampling_rate, std::ptr::null(), n_samples as u64, callback_len) }.max(1);
    let mut count = vec![0u32; n_streams];
    let mut idx = vec![0u32; n_streams * max_beats as usize];
    let mut ms = vec![0u32; n_streams * max_beats as usize];
    let rc = unsafe {
        bd_analyze_batch_host(
            kind, sampling_rate, std::ptr::null(), samples.as_ptr(), n_streams as u64, n_samples as u64,
            n_samples as u64, callback_len, std::ptr::null_mut(), std::ptr::null_mut(), count.as_mut_ptr(),
            idx.as_mut_ptr(), ms.as_mut_ptr(), max_beats, -1,
        )
    };
    if rc != BD_OK {
        return Err(last_error());
    }
    Ok((0..n_streams)
        .map(|s| {
            let n = (count[s] as usize).min(max_beats as usize);
            (0..n).map(|i| (idx[s * max_beats as usize + i], ms[s * max_beats as usize + i])).collect()
        })
        .collect())
}
