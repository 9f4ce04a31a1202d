//! `record::start_listening` (`src/record.rs:39-149` of the parent crate) over
//! `bd_start_listening`: a thread feeds every callback of an input to a detector
//! and calls `on_beat_cb` for every beat until `keep_recording` becomes false.
//!
//! The parent opens a `cpal::Device`; this crate has no audio dependency, so the
//! input is anything that implements [`InputDevice`] (a cpal stream adapter, a
//! file reader, a test source).  Everything else — argument order, the error
//! strings, `Arc<AtomicBool>` as the stop flag, `JoinHandle<()>` as the result,
//! `on_beat_cb` running on the listening thread — is the parent's contract.
//! A parent crate that keeps its own `record.rs` needs none of this: that file
//! only calls `StrategyKind::detector` and `Strategy::is_beat`.
use crate::ffi;
use crate::{last_error, BeatInfo, StrategyKind};
use std::os::raw::{c_int, c_void};
use std::sync::atomic::{AtomicBool, Ordering};
use std::sync::Arc;
use std::thread::{spawn, JoinHandle};

/// Sample formats of an input (`cpal::SampleFormat`, `src/record.rs:90-133`).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum SampleFormat {
    I16,
    U16,
    F32,
}

impl SampleFormat {
    pub const fn as_raw(&self) -> c_int {
        match self {
            SampleFormat::I16 => ffi::BD_FMT_I16,
            SampleFormat::U16 => ffi::BD_FMT_U16,
            SampleFormat::F32 => ffi::BD_FMT_F32,
        }
    }

    pub const fn bytes_per_sample(&self) -> usize {
        match self {
            SampleFormat::F32 => 4,
            _ => 2,
        }
    }
}

/// What `start_listening` needs from an input device.
pub trait InputDevice: Send + 'static {
    /// `default_input_config().sample_rate()`, `src/record.rs:58`.
    fn sampling_rate(&self) -> u32;
    /// `default_input_config().sample_format()`, `src/record.rs:59`.
    fn sample_format(&self) -> SampleFormat;
    /// Longest callback the device delivers, in samples.
    fn max_callback_len(&self) -> u32 {
        4096
    }
    /// Blocks until the next callback is available, writes its samples (native
    /// endianness, `sample_format()`) to the front of `buffer` and returns their
    /// number; 0 ends the stream.
    fn read(&mut self, buffer: &mut [u8]) -> usize;
}

/// What the two trampolines share.  Both only ever form a shared reference to it: the library's
/// listening thread reads `keep_flag` through a raw pointer while they run, so no `&mut Context`
/// may exist.  The one field that needs exclusive access, the device, sits in an `UnsafeCell`;
/// only `read_trampoline` touches it, and the library calls the trampolines from its one listening
/// thread, one after the other.
struct Context<F: Fn(BeatInfo) + Send + 'static, D: InputDevice> {
    on_beat: F,
    device: std::cell::UnsafeCell<D>,
    bytes_per_sample: usize,
    keep: Arc<AtomicBool>,
    // the C side polls an int: the read trampoline mirrors the AtomicBool into it before every callback
    keep_flag: std::sync::atomic::AtomicI32,
}

unsafe extern "C" fn on_beat_trampoline<F: Fn(BeatInfo) + Send + 'static, D: InputDevice>(user: *mut c_void, relative_ms: u32) {
    let ctx = &*(user as *const Context<F, D>);
    (ctx.on_beat)(BeatInfo::new(relative_ms));
}

unsafe extern "C" fn read_trampoline<F: Fn(BeatInfo) + Send + 'static, D: InputDevice>(user: *mut c_void, buffer: *mut c_void, capacity: usize) -> usize {
    let ctx = &*(user as *const Context<F, D>);
    // the stop flag is sampled once per callback (`src/record.rs:141-145` polls it in a loop)
    if !ctx.keep.load(Ordering::SeqCst) {
        ctx.keep_flag.store(0, Ordering::SeqCst);
        return 0;
    }
    let bytes = std::slice::from_raw_parts_mut(buffer as *mut u8, capacity * ctx.bytes_per_sample);
    // exclusive by construction: nothing else reads or writes the device (see `Context`)
    let device = &mut *ctx.device.get();
    device.read(bytes).min(capacity)
}

/// Starts listening on `input_dev` and calls `on_beat_cb` on each recognised
/// beat, on a new thread (`src/record.rs:39-44`).
///
/// Errors are the parent's: `Err("Variable keep_recording is false from the
/// beginning!?")` and `Err("Must have input device!")` when there is no input
/// (`src/record.rs:45-47,51-55`; without an audio library there is no default
/// device to fall back to).  A failure inside the thread panics there, which
/// `JoinHandle::join` reports, as in the parent.
pub fn start_listening<D: InputDevice>(
    on_beat_cb: impl Fn(BeatInfo) + Send + 'static,
    input_dev: Option<D>,
    strategy: StrategyKind,
    keep_recording: Arc<AtomicBool>,
) -> Result<JoinHandle<()>, String> {
    if !keep_recording.load(Ordering::SeqCst) {
        return Err("Variable keep_recording is false from the beginning!?".to_string());
    }
    let device = input_dev.ok_or_else(|| "Must have input device!".to_string())?;
    start(on_beat_cb, device, strategy, keep_recording)
}

fn start<F: Fn(BeatInfo) + Send + 'static, D: InputDevice>(
    on_beat_cb: F,
    device: D,
    strategy: StrategyKind,
    keep_recording: Arc<AtomicBool>,
) -> Result<JoinHandle<()>, String> {
    let format = device.sample_format();
    let source_header = (device.sampling_rate(), format.as_raw(), device.max_callback_len());
    eprintln!("  sampling_rate: {}", source_header.0);
    eprintln!("  sample_format: {:?}", format);
    let ctx = Box::new(Context {
        on_beat: on_beat_cb,
        device: std::cell::UnsafeCell::new(device),
        bytes_per_sample: format.bytes_per_sample(),
        keep: keep_recording,
        keep_flag: std::sync::atomic::AtomicI32::new(1),
    });
    let ctx_ptr = Box::into_raw(ctx);
    let source = ffi::bd_input_source {
        sampling_rate: source_header.0,
        sample_format: source_header.1,
        max_callback_len: source_header.2,
        read: Some(read_trampoline::<F, D>),
        user: ctx_ptr as *mut c_void,
    };
    let mut listener: *mut ffi::bd_listener = std::ptr::null_mut();
    let rc = unsafe {
        ffi::bd_start_listening(
            Some(on_beat_trampoline::<F, D>),
            ctx_ptr as *mut c_void,
            &source,
            strategy.as_raw(),
            // (field address taken without forming a reference to the whole context)
            std::ptr::addr_of!((*ctx_ptr).keep_flag) as *const c_int,
            &mut listener,
        )
    };
    if rc != ffi::BD_OK {
        unsafe { drop(Box::from_raw(ctx_ptr)) };
        return Err(last_error());
    }
    // the library's thread does the listening; this one owns the context and turns the
    // listener's status into the thread result a `JoinHandle` carries
    struct SendPtr<T>(*mut T);
    unsafe impl<T> Send for SendPtr<T> {}
    let (listener, ctx_ptr) = (SendPtr(listener), SendPtr(ctx_ptr));
    Ok(spawn(move || {
        let (listener, ctx_ptr) = (listener, ctx_ptr);
        let status = unsafe { ffi::bd_listener_join(listener.0) };
        unsafe { drop(Box::from_raw(ctx_ptr.0)) };
        if status != ffi::BD_OK {
            panic!("listening thread failed: status {} ({})", status, last_error());
        }
    }))
}

/// `record::u16_data_to_i16` (`src/record.rs:151-158`) on the host: the offset is
/// `i16::MAX / 2`, and the narrowing wraps.
#[inline(always)]
pub fn u16_data_to_i16(data: &[u16]) -> Vec<i16> {
    let mut out = Vec::with_capacity(data.len());
    for &x in data {
        out.push((x as i32 - (i16::MAX / 2) as i32) as i16);
    }
    out
}

/// `record::f32_data_to_i16` (`src/record.rs:160-166`) on the host: scale by
/// `i16::MAX`, then the saturating `as` cast (NaN becomes 0).
#[inline(always)]
pub fn f32_data_to_i16(data: &[f32]) -> Vec<i16> {
    let mut out = Vec::with_capacity(data.len());
    for &x in data {
        out.push((x * i16::MAX as f32) as i16);
    }
    out
}

/// GPUs this process can use: what `audio_input_device_list` is to inputs
/// (`src/record.rs:170-177`), for the side the work runs on.
pub fn gpu_count() -> usize {
    let mut n: c_int = 0;
    unsafe { ffi::bd_device_count(&mut n) };
    n.max(0) as usize
}

/// Name -> index of every GPU, ordered by name like the reference's
/// `audio_input_device_list` (`src/record.rs:170-177`) orders its cpal devices.
pub fn gpu_device_list() -> std::collections::BTreeMap<String, usize> {
    let mut map = std::collections::BTreeMap::new();
    for i in 0..gpu_count() {
        let mut buf = [0 as std::os::raw::c_char; 256];
        let status = unsafe { ffi::bd_device_name(i as c_int, buf.as_mut_ptr(), buf.len(), std::ptr::null_mut()) };
        let name = if status == 0 {
            unsafe { std::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()
        } else {
            format!("Unknown device #{}", i)
        };
        map.insert(format!("[{}] {}", i, name), i);
    }
    map
}
