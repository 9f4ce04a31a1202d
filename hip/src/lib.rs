//! GPU (MI355X / gfx950) implementation of beat-detector's per-window analysis
//! path behind the crate's own public interface: [`BeatInfo`], [`Strategy`],
//! [`StrategyKind`] and [`record::start_listening`] have the shapes of the
//! parent crate (`src/lib.rs:51-185`, `src/record.rs:39-44`), so a caller written
//! against it compiles against this crate by changing the `use` line.  All
//! arithmetic happens in `libbeatdet_hip.so` (`include/beat_detector_hip.h`),
//! declared in [`ffi`]; there is no CPU implementation here.
//!
//! SOURCE ONLY: the environment this repository is built in has no Rust
//! toolchain, so the crate has not been compiled.  `tests/test_cabi_symbols.py`
//! keeps every declaration of [`ffi`] equal, argument by argument, to the header.
//!
//! Two ways to wire it into the parent crate:
//! * replace the bodies of `StrategyKind::detector/name/description` by the ones
//!   below (three lines each) and leave `record.rs` untouched: it only calls
//!   `strategy.detector(rate)` and `detector.is_beat(&[i16])`
//!   (`src/record.rs:85,95,109,123`);
//! * or depend on this crate and re-export `BeatInfo`, `Strategy`, `StrategyKind`.
pub mod ffi;
pub mod record;

use std::ffi::CStr;
use std::os::raw::c_int;

/// Message of the last failed call on this thread (`bd_last_error`).
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::bd_last_error()).to_string_lossy().into_owned() }
}

fn static_str(p: *const std::os::raw::c_char) -> &'static str {
    // bd_kind_name / bd_kind_description return string literals of the library
    if p.is_null() {
        return "";
    }
    unsafe { CStr::from_ptr(p) }.to_str().unwrap_or("")
}

/// Information about a detected beat (`src/lib.rs:51-66`).
#[derive(Debug)]
pub struct BeatInfo {
    relative_ms: u32,
}

impl BeatInfo {
    #[inline(always)]
    pub const fn new(relative_ms: u32) -> Self {
        Self { relative_ms }
    }

    #[inline(always)]
    pub const fn relative_ms(&self) -> u32 {
        self.relative_ms
    }
}

/// The per-window strategy interface (`src/lib.rs:72-137`).  The two shared
/// gates of the parent trait (`last_beat_beyond_threshold`,
/// `amplitude_high_enough`) take the parent's private `AnalysisState` /
/// `WindowStats`, so they cannot be provided methods here; on the GPU they are
/// part of the kernel.  For strategies ported from the parent crate they exist
/// as the free functions [`last_beat_beyond_threshold`] and
/// [`amplitude_high_enough`] over the plain numbers those types hold.
pub trait Strategy {
    /// `Some(BeatInfo)` if the callback's samples complete a beat.
    fn is_beat(&self, samples: &[i16]) -> Option<BeatInfo>;

    /// The [`StrategyKind`] of this detector.
    fn kind(&self) -> StrategyKind;

    fn name() -> &'static str
    where
        Self: Sized;

    fn description() -> &'static str
    where
        Self: Sized;

    fn min_duration_between_beats_ms() -> u32
    where
        Self: Sized;
}

/// `Strategy::last_beat_beyond_threshold` (`src/lib.rs:115-127`) over the numbers an
/// `AnalysisState` holds: `beat_time_ms` is the middle of the current callback on the
/// detector's clock, `last_beat_ms` the time of the last beat (0 before the first),
/// `min_duration_ms` the strategy's `min_duration_between_beats_ms()`.  `false` means the
/// callback comes too soon after the last beat and cannot be one.
#[inline(always)]
pub const fn last_beat_beyond_threshold(beat_time_ms: u32, last_beat_ms: u32, min_duration_ms: u32) -> bool {
    // only check this if at least a single beat was recognized
    if beat_time_ms > 0 {
        let threshold = last_beat_ms + min_duration_ms;
        if beat_time_ms < threshold {
            return false;
        }
    }
    true
}

/// `Strategy::amplitude_high_enough` (`src/lib.rs:133-136`) over the number a `WindowStats`
/// holds: the largest absolute sample of the callback (`window_stats.rs:41-55`).
#[inline(always)]
pub const fn amplitude_high_enough(max_amplitude: u16) -> bool {
    const MIN_AMPLITUDE_THRESHOLD: i16 = (i16::MAX as f32 * 0.3) as i16;
    max_amplitude >= MIN_AMPLITUDE_THRESHOLD as u16
}

/// `src/lib.rs:142-185`; discriminants follow `values()` and the C header.
#[derive(Debug, PartialEq, Eq, Hash, Clone, Copy)]
#[non_exhaustive]
pub enum StrategyKind {
    LPF,
    Spectrum,
}

impl StrategyKind {
    /// The `kind` integer of the C ABI (`BD_KIND_*`).
    #[inline(always)]
    pub const fn as_raw(&self) -> c_int {
        match self {
            StrategyKind::LPF => ffi::BD_KIND_LPF,
            StrategyKind::Spectrum => ffi::BD_KIND_SPECTRUM,
        }
    }

    /// A detector whose state and arithmetic live on the GPU
    /// (`StrategyKind::detector`, `src/lib.rs:155-161`).  Panics if the library
    /// cannot create it (no GPU): the parent's constructors cannot fail, and its
    /// set-up errors are panics as well (`src/record.rs:56,135,138`).
    #[inline(always)]
    pub fn detector(&self, sampling_rate: u32) -> Box<dyn Strategy + Send> {
        match self {
            StrategyKind::LPF => Box::new(HipLpfDetector(HipDetector::new(*self, sampling_rate))),
            StrategyKind::Spectrum => Box::new(HipSpectrumDetector(HipDetector::new(*self, sampling_rate))),
        }
    }

    pub fn name(&self) -> &'static str {
        static_str(unsafe { ffi::bd_kind_name(self.as_raw()) })
    }

    pub fn description(&self) -> &'static str {
        static_str(unsafe { ffi::bd_kind_description(self.as_raw()) })
    }

    pub fn values() -> Vec<Self> {
        vec![Self::LPF, Self::Spectrum]
    }
}

/// Owner of one `bd_detector` handle.
#[derive(Debug)]
pub struct HipDetector {
    raw: *mut ffi::bd_detector,
    kind: StrategyKind,
}

// The parent's detectors are `Send` but not `Sync` (`Box<dyn Strategy + Send>`, src/lib.rs:155);
// a handle is used by one thread at a time.
unsafe impl Send for HipDetector {}

impl HipDetector {
    pub fn try_new(kind: StrategyKind, sampling_rate: u32) -> Result<Self, String> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { ffi::bd_detector_create(kind.as_raw(), sampling_rate, std::ptr::null(), -1, &mut raw) };
        if rc == ffi::BD_OK {
            Ok(Self { raw, kind })
        } else {
            Err(last_error())
        }
    }

    pub fn new(kind: StrategyKind, sampling_rate: u32) -> Self {
        match Self::try_new(kind, sampling_rate) {
            Ok(d) => d,
            Err(e) => panic!("bd_detector_create: {}", e),
        }
    }

    /// `Strategy::is_beat`.  An empty slice panics like the parent (`window_stats.rs:51`).
    pub fn is_beat(&self, samples: &[i16]) -> Option<BeatInfo> {
        let mut ms = 0u32;
        match unsafe { ffi::bd_detector_is_beat(self.raw, samples.as_ptr(), samples.len(), &mut ms) } {
            1 => Some(BeatInfo::new(ms)),
            0 => None,
            e => panic!("bd_detector_is_beat: status {} ({})", e, last_error()),
        }
    }

    pub fn kind(&self) -> StrategyKind {
        self.kind
    }

    /// (analysed value, decision before the refractory rule) of the last call.
    pub fn last(&self) -> (f32, bool) {
        let (mut v, mut c) = (0f32, 0 as c_int);
        unsafe { ffi::bd_detector_last(self.raw, &mut v, &mut c) };
        (v, c != 0)
    }
}

impl Drop for HipDetector {
    fn drop(&mut self) {
        unsafe { ffi::bd_detector_destroy(self.raw) }
    }
}

/// Replacement of `LpfBeatDetector` (`src/strategies/lpf.rs:34-115`).
#[derive(Debug)]
pub struct HipLpfDetector(HipDetector);

/// Replacement of `SABeatDetector` (`src/strategies/spectrum.rs:37-132`).
#[derive(Debug)]
pub struct HipSpectrumDetector(HipDetector);

impl HipLpfDetector {
    pub fn new(sampling_rate: u32) -> Self {
        Self(HipDetector::new(StrategyKind::LPF, sampling_rate))
    }
}

impl HipSpectrumDetector {
    pub fn new(sampling_rate: u32) -> Self {
        Self(HipDetector::new(StrategyKind::Spectrum, sampling_rate))
    }
}

impl Strategy for HipLpfDetector {
    #[inline(always)]
    fn is_beat(&self, samples: &[i16]) -> Option<BeatInfo> {
        self.0.is_beat(samples)
    }

    fn kind(&self) -> StrategyKind {
        StrategyKind::LPF
    }

    fn name() -> &'static str {
        StrategyKind::LPF.name()
    }

    fn description() -> &'static str {
        StrategyKind::LPF.description()
    }

    fn min_duration_between_beats_ms() -> u32 {
        unsafe { ffi::bd_kind_min_duration_between_beats_ms(ffi::BD_KIND_LPF) }
    }
}

impl Strategy for HipSpectrumDetector {
    #[inline(always)]
    fn is_beat(&self, samples: &[i16]) -> Option<BeatInfo> {
        self.0.is_beat(samples)
    }

    fn kind(&self) -> StrategyKind {
        StrategyKind::Spectrum
    }

    fn name() -> &'static str {
        StrategyKind::Spectrum.name()
    }

    fn description() -> &'static str {
        StrategyKind::Spectrum.description()
    }

    fn min_duration_between_beats_ms() -> u32 {
        unsafe { ffi::bd_kind_min_duration_between_beats_ms(ffi::BD_KIND_SPECTRUM) }
    }
}

/// Result of a batched analysis: per stream, `(callback index, BeatInfo)`.
pub type BeatLists = Vec<Vec<(u32, BeatInfo)>>;

fn beat_lists(count: &[u32], idx: &[u32], ms: &[u32], max_beats: usize) -> BeatLists {
    count
        .iter()
        .enumerate()
        .map(|(s, &c)| {
            let n = (c as usize).min(max_beats);
            (0..n).map(|i| (idx[s * max_beats + i], BeatInfo::new(ms[s * max_beats + i]))).collect()
        })
        .collect()
}

/// The new batched entry point: many equally long streams (stream-major
/// `samples`, `n_samples` each) through one strategy on the current GPU; for
/// every stream the result of the loop `for window in chunks(callback_len) {
/// detector.is_beat(window) }` of the parent's harness (`src/lib.rs:276-292`).
pub fn analyze_batch(kind: StrategyKind, sampling_rate: u32, samples: &[i16], n_samples: usize, callback_len: u32) -> Result<BeatLists, String> {
    if n_samples == 0 || samples.len() % n_samples != 0 {
        return Err("samples.len() must be a multiple of n_samples".to_string());
    }
    let n_streams = samples.len() / n_samples;
    let max_beats =
        unsafe { ffi::bd_batch_max_beats(kind.as_raw() as u32, sampling_rate, std::ptr::null(), n_samples as u64, callback_len) }.max(1);
    let mut count = vec![0u32; n_streams];
    let mut idx = vec![0u32; n_streams * max_beats as usize];
    let mut ms = vec![0u32; n_streams * max_beats as usize];
    let rc = unsafe {
        ffi::bd_analyze_batch_host(
            kind.as_raw(),
            sampling_rate,
            std::ptr::null(),
            samples.as_ptr(),
            n_streams as u64,
            n_samples as u64,
            n_samples as u64,
            callback_len,
            std::ptr::null_mut(),
            std::ptr::null_mut(),
            count.as_mut_ptr(),
            idx.as_mut_ptr(),
            ms.as_mut_ptr(),
            max_beats,
            -1,
        )
    };
    if rc != ffi::BD_OK {
        return Err(last_error());
    }
    Ok(beat_lists(&count, &idx, &ms, max_beats as usize))
}

/// The batched entry point on memory that already lies on GPU `device`: `samples` is a device
/// pointer to `n_streams` rows of `stream_stride` samples, `flags` a device buffer of `n_streams`
/// rows of `n_samples / callback_len` bytes (the per-callback decisions, scratch for the caller),
/// `beat_count` / `beat_idx` / `beat_ms` device arrays (`[n_streams]`, `[n_streams][max_beats]`).
/// Only enqueues on `hip_stream` (null: the default stream) and allocates nothing: this is the
/// entry for throughput (the kernels read at HBM speed), and the one a multi-GPU program calls from
/// one thread or process per GPU with its own shard of the streams -- streams are independent, so no
/// data crosses GPUs.  [`analyze_batch`] and [`analyze_batch_multi`] take host slices instead and are
/// bounded by the copy over PCIe (about 30 GB/s per GPU, two orders of magnitude below the kernels).
///
/// # Safety
/// Every pointer must be a device pointer of `device` with the extent described above, valid until
/// the work enqueued on `hip_stream` has finished.
#[allow(clippy::too_many_arguments)]
pub unsafe fn analyze_batch_on(
    device: i32,
    kind: StrategyKind,
    sampling_rate: u32,
    samples: *const i16,
    n_streams: u64,
    n_samples: u64,
    stream_stride: u64,
    callback_len: u32,
    flags: *mut u8,
    beat_count: *mut u32,
    beat_idx: *mut u32,
    beat_ms: *mut u32,
    max_beats: u32,
    hip_stream: *mut std::os::raw::c_void,
) -> Result<(), String> {
    let mut prev: c_int = -1;
    if ffi::bd_current_device(&mut prev) != ffi::BD_OK || ffi::bd_set_device(device as c_int) != ffi::BD_OK {
        return Err(last_error());
    }
    let rc = ffi::bd_analyze_batch(
        kind.as_raw(),
        sampling_rate,
        std::ptr::null(),
        samples,
        n_streams,
        n_samples,
        stream_stride,
        callback_len,
        std::ptr::null_mut(),
        flags,
        0,
        beat_count,
        beat_idx,
        beat_ms,
        max_beats,
        hip_stream,
    );
    let message = if rc != ffi::BD_OK { last_error() } else { String::new() };
    if prev >= 0 && prev != device as c_int {
        ffi::bd_set_device(prev);
    }
    if rc != ffi::BD_OK {
        return Err(message);
    }
    Ok(())
}

/// One resident shard of a batch on one GPU: device pointers of `device`, as in [`analyze_batch_on`].
#[derive(Debug, Clone, Copy)]
pub struct DeviceShard {
    pub device: i32,
    pub samples: *const i16,
    pub n_streams: u64,
    pub flags: *mut u8,
    pub beat_count: *mut u32,
    pub beat_idx: *mut u32,
    pub beat_ms: *mut u32,
}

/// Every shard on its own GPU at once, from one call: the library starts one host thread per shard,
/// each selects its device, enqueues the batch on a stream of its own and waits for it.  Samples stay
/// where they are (no copy, no traffic between GPUs); the call returns when all shards have finished.
/// This is the multi-GPU entry for samples that are already resident; [`analyze_batch_multi`] is the
/// one for host slices.
///
/// # Safety
/// Every pointer of every shard must be a device pointer of that shard's `device` with the extents of
/// [`analyze_batch_on`], valid until the call returns.
pub unsafe fn analyze_batch_devices(
    kind: StrategyKind,
    sampling_rate: u32,
    shards: &[DeviceShard],
    n_samples: u64,
    stream_stride: u64,
    callback_len: u32,
    max_beats: u32,
) -> Result<(), String> {
    let mut samples: Vec<*const i16> = shards.iter().map(|s| s.samples).collect();
    let n_streams: Vec<u64> = shards.iter().map(|s| s.n_streams).collect();
    let mut flags: Vec<*mut u8> = shards.iter().map(|s| s.flags).collect();
    let mut count: Vec<*mut u32> = shards.iter().map(|s| s.beat_count).collect();
    let mut idx: Vec<*mut u32> = shards.iter().map(|s| s.beat_idx).collect();
    let mut ms: Vec<*mut u32> = shards.iter().map(|s| s.beat_ms).collect();
    let devices: Vec<c_int> = shards.iter().map(|s| s.device as c_int).collect();
    let rc = ffi::bd_analyze_batch_devices(
        kind.as_raw(),
        sampling_rate,
        std::ptr::null(),
        samples.as_mut_ptr(),
        n_streams.as_ptr(),
        n_samples,
        stream_stride,
        callback_len,
        std::ptr::null_mut(),
        flags.as_mut_ptr(),
        0,
        count.as_mut_ptr(),
        idx.as_mut_ptr(),
        ms.as_mut_ptr(),
        max_beats,
        devices.as_ptr(),
        devices.len() as c_int,
    );
    if rc != ffi::BD_OK {
        return Err(last_error());
    }
    Ok(())
}

/// The host-slice entry sharded by stream over several GPUs of the node (contiguous shards, no
/// traffic between GPUs).  Like [`analyze_batch`] it copies the samples over PCIe, which bounds
/// it at about 30 GB/s per GPU; with samples already on the GPUs use [`analyze_batch_devices`] (or
/// [`analyze_batch_on`] from a thread or process per device) instead.
pub fn analyze_batch_multi(kind: StrategyKind, sampling_rate: u32, samples: &[i16], n_samples: usize, callback_len: u32, devices: &[i32]) -> Result<BeatLists, String> {
    if n_samples == 0 || samples.len() % n_samples != 0 {
        return Err("samples.len() must be a multiple of n_samples".to_string());
    }
    let n_streams = samples.len() / n_samples;
    let max_beats =
        unsafe { ffi::bd_batch_max_beats(kind.as_raw() as u32, sampling_rate, std::ptr::null(), n_samples as u64, callback_len) }.max(1);
    let mut count = vec![0u32; n_streams];
    let mut idx = vec![0u32; n_streams * max_beats as usize];
    let mut ms = vec![0u32; n_streams * max_beats as usize];
    let devs: Vec<c_int> = devices.iter().map(|&d| d as c_int).collect();
    let rc = unsafe {
        ffi::bd_analyze_batch_multi(
            kind.as_raw(),
            sampling_rate,
            std::ptr::null(),
            samples.as_ptr(),
            n_streams as u64,
            n_samples as u64,
            n_samples as u64,
            callback_len,
            std::ptr::null_mut(),
            std::ptr::null_mut(),
            count.as_mut_ptr(),
            idx.as_mut_ptr(),
            ms.as_mut_ptr(),
            max_beats,
            devs.as_ptr(),
            devs.len() as c_int,
        )
    };
    if rc != ffi::BD_OK {
        return Err(last_error());
    }
    Ok(beat_lists(&count, &idx, &ms, max_beats as usize))
}

/// Many concurrent input streams advanced together, one callback per stream and
/// tick (`bd_live_*`): the multi-stream form of the loop in `src/record.rs:87-146`.
#[derive(Debug)]
pub struct Live {
    raw: *mut ffi::bd_live,
    n_streams: usize,
}

unsafe impl Send for Live {}

impl Live {
    pub fn new(kind: StrategyKind, sampling_rate: u32, n_streams: u32, max_callback_len: u32) -> Result<Self, String> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { ffi::bd_live_create(kind.as_raw(), sampling_rate, std::ptr::null(), n_streams, max_callback_len, 0, -1, &mut raw) };
        if rc == ffi::BD_OK {
            Ok(Self { raw, n_streams: n_streams as usize })
        } else {
            Err(last_error())
        }
    }

    /// One tick of `i16` samples, `[n_streams][n]`; returns the beat of every stream.
    pub fn tick(&mut self, samples: &[i16]) -> Result<Vec<Option<BeatInfo>>, String> {
        if self.n_streams == 0 || samples.len() % self.n_streams != 0 {
            return Err("samples.len() must be a multiple of the number of streams".to_string());
        }
        let n = (samples.len() / self.n_streams) as u32;
        let mut tick = 0u64;
        let rc = unsafe { ffi::bd_live_submit(self.raw, samples.as_ptr() as *const _, n, ffi::BD_FMT_I16, &mut tick) };
        if rc != ffi::BD_OK {
            return Err(last_error());
        }
        let mut beats = vec![0u8; self.n_streams];
        let mut ms = vec![0u32; self.n_streams];
        let rc = unsafe { ffi::bd_live_poll(self.raw, tick, 1, beats.as_mut_ptr(), ms.as_mut_ptr(), std::ptr::null_mut()) };
        if rc < 0 {
            return Err(last_error());
        }
        Ok(beats.iter().zip(ms.iter()).map(|(&b, &m)| if b != 0 { Some(BeatInfo::new(m)) } else { None }).collect())
    }
}

impl Live {
    /// One tick in which every stream brings its own number of `i16` samples
    /// (`bd_live_submit_ragged`): in the parent every stream is a device of its own whose
    /// closure gets its own `data.len()` (`src/record.rs:93-95,107-109,121-123`).  `samples`
    /// is `[n_streams][row_pitch]`, stream `m`'s callback the first `lengths[m]` samples of
    /// its row; a stream with length 0 skips the tick (its detector does not advance).
    pub fn tick_ragged(&mut self, samples: &[i16], lengths: &[u32]) -> Result<Vec<Option<BeatInfo>>, String> {
        if self.n_streams == 0 || lengths.len() != self.n_streams || samples.len() % self.n_streams != 0 {
            return Err("samples must be [n_streams][row_pitch], lengths [n_streams]".to_string());
        }
        let pitch = (samples.len() / self.n_streams) as u32;
        let mut tick = 0u64;
        let rc = unsafe {
            ffi::bd_live_submit_ragged(self.raw, samples.as_ptr() as *const _, lengths.as_ptr(), pitch, ffi::BD_FMT_I16, &mut tick)
        };
        if rc != ffi::BD_OK {
            return Err(last_error());
        }
        let mut beats = vec![0u8; self.n_streams];
        let mut ms = vec![0u32; self.n_streams];
        let rc = unsafe { ffi::bd_live_poll(self.raw, tick, 1, beats.as_mut_ptr(), ms.as_mut_ptr(), std::ptr::null_mut()) };
        if rc < 0 {
            return Err(last_error());
        }
        Ok(beats.iter().zip(ms.iter()).map(|(&b, &m)| if b != 0 { Some(BeatInfo::new(m)) } else { None }).collect())
    }
}

impl Drop for Live {
    fn drop(&mut self) {
        unsafe { ffi::bd_live_destroy(self.raw) }
    }
}
