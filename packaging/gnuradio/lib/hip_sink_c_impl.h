/* NOT COMPILED in the build container (no GNU Radio). */
#pragma once

#include <gnuradio/fosphor/hip_sink_c.h>

#include "fosphor_amd_sink.h"

namespace gr {
namespace fosphor {

class hip_sink_c_impl : public hip_sink_c
{
private:
    ::fosphor_amd::sink_runtime d_rt;      /* fifo + worker thread + settings state machine */
    double d_center, d_span;

public:
    hip_sink_c_impl();
    ~hip_sink_c_impl() override;

    /* base_sink_c */
    void execute_ui_action(enum ui_action_t action) override;
    void execute_mouse_action(enum mouse_action_t action, int x, int y) override;
    void set_frequency_range(const double center, const double span) override;
    void set_frequency_center(const double center) override;
    void set_frequency_span(const double span) override;
    void set_fft_window(const gr::fft::window::win_type win) override;

    /* hip_sink_c */
    struct ::fosphor* core() override { return d_rt.core(); }
    uint64_t frames() const override { return d_rt.frames(); }
    uint64_t samples_processed() const override { return d_rt.samples_processed(); }

    /* gr::sync_block */
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override;
    bool start() override;
    bool stop() override;
};

} // namespace fosphor
} // namespace gr
