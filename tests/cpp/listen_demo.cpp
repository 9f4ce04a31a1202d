// A caller written against the C++ mirror of the reference interface: the shape
// of the reference's examples/minimal.rs (select a strategy, start_listening
// with an on_beat closure, join) with a synthetic input device in place of cpal.
// Prints "beat <relative_ms>" per beat; used by tests/test_cpp_mirror.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "beat_detector.hpp"

using namespace beat_detector;

int main(int argc, char **argv)
{
    const int kind_i = argc > 1 ? std::atoi(argv[1]) : 1;
    const uint32_t seconds = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 5;
    const uint64_t stream_id = argc > 3 ? (uint64_t)std::atoll(argv[3]) : 7;
    const uint32_t sr = 44100;
    const StrategyKind strategy = kind_i == 0 ? StrategyKind::LPF : StrategyKind::Spectrum;
    std::printf("strategy %s\n", name(strategy));
    for (StrategyKind k : values()) std::printf("kind %d: %s (%u ms)\n", (int)k, name(k), min_duration_between_beats_ms(k));

    std::vector<int16_t> audio((size_t)sr * seconds);
    if (bd_synth_fill_host(audio.data(), 1, audio.size(), audio.size(), stream_id, 123, sr) != BD_OK) return 2;

    record::InputDevice dev;
    dev.sampling_rate = sr;
    dev.sample_format = BD_FMT_I16;
    size_t pos = 0;
    dev.read = [&](void *buffer, size_t capacity) -> size_t {
        const size_t n = std::min<size_t>({(size_t)1024, capacity, audio.size() - pos}); // fixed 1024 (record.rs:71-77)
        std::memcpy(buffer, audio.data() + pos, n * sizeof(int16_t));
        pos += n;
        return n;
    };
    auto recording = std::make_shared<std::atomic<int>>(1);

    // keep_recording false from the beginning is an Err (record.rs:45-47)
    auto off = std::make_shared<std::atomic<int>>(0);
    auto bad = record::start_listening([](BeatInfo) {}, dev, strategy, off);
    if (bad.is_ok()) return 3;
    std::printf("err: %s\n", bad.error.c_str());

    auto handle = record::start_listening([](BeatInfo info) { std::printf("beat %u\n", info.relative_ms()); },
                                          dev, strategy, recording);
    if (!handle.is_ok()) { std::fprintf(stderr, "start_listening: %s\n", handle.error.c_str()); return 1; }
    auto joined = handle.value->join();
    if (!joined.is_ok()) { std::fprintf(stderr, "join: %s\n", joined.error.c_str()); return 1; }

    // the per-window API directly
    auto det = detector(strategy, sr);
    size_t beats = 0;
    for (size_t p = 0; p + 1024 <= audio.size(); p += 1024)
        if (det->is_beat(audio.data() + p, 1024)) beats++;
    std::printf("detector beats %zu\n", beats);

    // two streams through the live engine, each with its own callback lengths in every tick: stream 0 is the
    // same audio in callbacks of 1024, stream 1 the same audio in callbacks of 512 (two ticks per callback of
    // stream 0, which skips every other tick)
    Live live(strategy, sr, 2, 1024);
    std::vector<int16_t> rows(2 * 1024);
    size_t beats0 = 0, beats1 = 0, p0 = 0, p1 = 0;
    for (size_t t = 0; p1 + 512 <= audio.size(); t++) {
        uint32_t len[2] = {0, 512};
        if (t % 2 == 0 && p0 + 1024 <= audio.size()) { std::memcpy(rows.data(), audio.data() + p0, 1024 * sizeof(int16_t)); len[0] = 1024; p0 += 1024; }
        std::memcpy(rows.data() + 1024, audio.data() + p1, 512 * sizeof(int16_t));
        p1 += 512;
        const auto r = live.tick_ragged(rows.data(), len, 1024);
        beats0 += r[0].has_value();
        beats1 += r[1].has_value();
    }
    std::printf("live ragged beats %zu %zu\n", beats0, beats1);
    return 0;
}
