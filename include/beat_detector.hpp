// beat_detector.hpp -- C++ host-side mirror of the reference crate's public
// interface for the per-window analysis path, over the C ABI of
// beat_detector_hip.h.  Header-only; link with -lbeatdet_hip.
//
//   reference (Rust)                               here
//   BeatInfo              src/lib.rs:51-66         beat_detector::BeatInfo
//   trait Strategy        src/lib.rs:72-137        beat_detector::Strategy
//   enum StrategyKind     src/lib.rs:142-185       beat_detector::StrategyKind + free functions
//   record::start_listening  src/record.rs:39-149  beat_detector::record::start_listening
//   record::{u16,f32}_data_to_i16  record.rs:151-166   done on the GPU inside the live engine
//
// Error behaviour follows the reference: set-up failures that the reference
// turns into Err(String) are returned as Result<...>::error; conditions on
// which the reference panics (empty window, unusable sampling rate) throw
// std::runtime_error.
#ifndef BEAT_DETECTOR_HPP
#define BEAT_DETECTOR_HPP

#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "beat_detector_hip.h"

namespace beat_detector {

/// src/lib.rs:51-66
class BeatInfo {
public:
    explicit constexpr BeatInfo(uint32_t relative_ms) : relative_ms_(relative_ms) {}
    constexpr uint32_t relative_ms() const { return relative_ms_; }
private:
    uint32_t relative_ms_;
};

/// src/lib.rs:142-149; discriminants follow StrategyKind::values()
enum class StrategyKind : int { LPF = BD_KIND_LPF, Spectrum = BD_KIND_SPECTRUM };

/// src/lib.rs:72-137.  Not thread-safe (the reference's detectors are Send, not Sync).
class Strategy {
public:
    virtual ~Strategy() = default;
    /// fn is_beat(&self, samples: &[i16]) -> Option<BeatInfo>, src/lib.rs:78
    virtual std::optional<BeatInfo> is_beat(const int16_t *samples, size_t n) = 0;
    std::optional<BeatInfo> is_beat(const std::vector<int16_t> &samples) { return is_beat(samples.data(), samples.size()); }
    /// src/lib.rs:82
    virtual StrategyKind kind() const = 0;
};

namespace detail {
inline std::string last_error() { return std::string(bd_last_error()); }

class HipDetector final : public Strategy {
public:
    HipDetector(StrategyKind kind, uint32_t sampling_rate, const bd_params *params, int device) : kind_(kind)
    {
        const int rc = bd_detector_create(static_cast<int>(kind), sampling_rate, params, device, &raw_);
        if (rc != BD_OK) throw std::runtime_error("bd_detector_create: " + last_error());
    }
    ~HipDetector() override { bd_detector_destroy(raw_); }
    HipDetector(const HipDetector &) = delete;
    HipDetector &operator=(const HipDetector &) = delete;

    std::optional<BeatInfo> is_beat(const int16_t *samples, size_t n) override
    {
        uint32_t ms = 0;
        const int rc = bd_detector_is_beat(raw_, samples, n, &ms);
        if (rc == 1) return BeatInfo(ms);
        if (rc == 0) return std::nullopt;
        throw std::runtime_error("bd_detector_is_beat: " + last_error()); // the reference panics (window_stats.rs:51)
    }
    StrategyKind kind() const override { return kind_; }

private:
    bd_detector *raw_ = nullptr;
    StrategyKind kind_;
};
} // namespace detail

/// StrategyKind::detector(&self, sampling_rate) -> Box<dyn Strategy + Send>, src/lib.rs:155-161
inline std::unique_ptr<Strategy> detector(StrategyKind kind, uint32_t sampling_rate,
                                          const bd_params *params = nullptr, int device = -1)
{
    return std::make_unique<detail::HipDetector>(kind, sampling_rate, params, device);
}
/// StrategyKind::name, src/lib.rs:164-170
inline const char *name(StrategyKind kind) { return bd_kind_name(static_cast<int>(kind)); }
/// StrategyKind::description, src/lib.rs:173-179
inline const char *description(StrategyKind kind) { return bd_kind_description(static_cast<int>(kind)); }
/// Strategy::min_duration_between_beats_ms, spectrum.rs:126-131 / lpf.rs:109-114
inline uint32_t min_duration_between_beats_ms(StrategyKind kind)
{
    return bd_kind_min_duration_between_beats_ms(static_cast<int>(kind));
}
/// StrategyKind::values, src/lib.rs:182-184
inline std::vector<StrategyKind> values() { return {StrategyKind::LPF, StrategyKind::Spectrum}; }

/// Many concurrent input streams advanced together, one callback per stream and tick (bd_live_*): the
/// multi-stream form of the loop in src/record.rs:87-146.  One thread at a time per engine.
class Live {
public:
    Live(StrategyKind kind, uint32_t sampling_rate, uint32_t n_streams, uint32_t max_callback_len,
         const bd_params *params = nullptr, int device = -1) : n_streams_(n_streams)
    {
        const int rc = bd_live_create(static_cast<int>(kind), sampling_rate, params, n_streams, max_callback_len, 0, device, &raw_);
        if (rc != BD_OK) throw std::runtime_error("bd_live_create: " + detail::last_error());
    }
    ~Live() { bd_live_destroy(raw_); }
    Live(const Live &) = delete;
    Live &operator=(const Live &) = delete;

    /// One tick of i16 samples [n_streams][n]: the beat of every stream.
    std::vector<std::optional<BeatInfo>> tick(const int16_t *samples, uint32_t n)
    {
        uint64_t id = 0;
        if (bd_live_submit(raw_, samples, n, BD_FMT_I16, &id) != BD_OK) throw std::runtime_error("bd_live_submit: " + detail::last_error());
        return collect(id);
    }
    /// One tick in which stream m brings lengths[m] samples, the first ones of row m of
    /// samples [n_streams][row_pitch]; 0 skips the stream (every stream is a device of its own in the
    /// reference and gets its own data.len(): src/record.rs:93-95, 107-109, 121-123).
    std::vector<std::optional<BeatInfo>> tick_ragged(const int16_t *samples, const uint32_t *lengths, uint32_t row_pitch)
    {
        uint64_t id = 0;
        if (bd_live_submit_ragged(raw_, samples, lengths, row_pitch, BD_FMT_I16, &id) != BD_OK)
            throw std::runtime_error("bd_live_submit_ragged: " + detail::last_error());
        return collect(id);
    }

private:
    std::vector<std::optional<BeatInfo>> collect(uint64_t id)
    {
        std::vector<uint8_t> beats(n_streams_);
        std::vector<uint32_t> ms(n_streams_);
        if (bd_live_poll(raw_, id, 1, beats.data(), ms.data(), nullptr) < 0) throw std::runtime_error("bd_live_poll: " + detail::last_error());
        std::vector<std::optional<BeatInfo>> out(n_streams_);
        for (uint32_t m = 0; m < n_streams_; m++)
            if (beats[m]) out[m] = BeatInfo(ms[m]);
        return out;
    }
    bd_live *raw_ = nullptr;
    uint32_t n_streams_;
};

/// Result<T, String>
template <class T>
struct Result {
    std::optional<T> value;
    std::string      error;
    bool is_ok() const { return value.has_value(); }
};

namespace record {

/// Stand-in for cpal::Device: something that delivers callbacks of samples.
struct InputDevice {
    uint32_t sampling_rate = 44100;
    int      sample_format = BD_FMT_I16; // cpal::SampleFormat, record.rs:90-133
    uint32_t max_callback_len = 4096;
    /// Blocks until the next callback, writes at most `capacity` samples of
    /// `sample_format` into `buffer`, returns their number (0 = stream ended).
    std::function<size_t(void *buffer, size_t capacity)> read;
};

/// JoinHandle<()>
class JoinHandle {
public:
    JoinHandle(bd_listener *h, std::shared_ptr<void> keep) : h_(h), keep_(std::move(keep)) {}
    JoinHandle(JoinHandle &&o) noexcept : h_(o.h_), keep_(std::move(o.keep_)) { o.h_ = nullptr; }
    JoinHandle(const JoinHandle &) = delete;
    ~JoinHandle() { if (h_) bd_listener_join(h_); }
    /// join(): Err if the listening thread failed
    Result<bool> join()
    {
        const int rc = h_ ? bd_listener_join(h_) : BD_OK;
        h_ = nullptr;
        if (rc != BD_OK) return {std::nullopt, detail::last_error()};
        return {true, {}};
    }
private:
    bd_listener          *h_;
    std::shared_ptr<void> keep_;
};

/// pub fn start_listening(on_beat_cb, input_dev, strategy, keep_recording)
///   -> Result<JoinHandle<()>, String>, src/record.rs:39-44.
/// `keep_recording` plays the Arc<AtomicBool>: non-zero = keep going.
inline Result<JoinHandle> start_listening(std::function<void(BeatInfo)> on_beat_cb,
                                          std::optional<InputDevice> input_dev, StrategyKind strategy,
                                          std::shared_ptr<std::atomic<int>> keep_recording)
{
    struct Ctx {
        std::function<void(BeatInfo)> cb;
        std::optional<InputDevice> dev;
        std::shared_ptr<std::atomic<int>> keep;
    };
    auto ctx = std::make_shared<Ctx>(Ctx{std::move(on_beat_cb), std::move(input_dev), std::move(keep_recording)});
    bd_input_source src{};
    const bd_input_source *psrc = nullptr;
    if (ctx->dev && ctx->dev->read) {
        src.sampling_rate = ctx->dev->sampling_rate;
        src.sample_format = ctx->dev->sample_format;
        src.max_callback_len = ctx->dev->max_callback_len;
        src.user = ctx.get();
        src.read = [](void *user, void *buffer, size_t capacity) -> size_t {
            return static_cast<Ctx *>(user)->dev->read(buffer, capacity);
        };
        psrc = &src;
    }
    static_assert(sizeof(std::atomic<int>) == sizeof(int), "atomic<int> must be layout-compatible with int");
    bd_listener *h = nullptr;
    const int rc = bd_start_listening(
        [](void *user, uint32_t ms) { static_cast<Ctx *>(user)->cb(BeatInfo(ms)); }, ctx.get(), psrc,
        static_cast<int>(strategy), reinterpret_cast<const volatile int *>(ctx->keep.get()), &h);
    if (rc != BD_OK) return {std::nullopt, detail::last_error()};
    return {JoinHandle(h, ctx), {}};
}

} // namespace record
} // namespace beat_detector

#endif // BEAT_DETECTOR_HPP
