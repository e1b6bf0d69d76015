/*
 * Headless fosphor sink on the HIP compute core.
 * NOT COMPILED in the build container (no GNU Radio); every call below lands in code that is
 * (fosphor_amd::sink_runtime, tests/test_gpu_parity.py::test_sink_runtime_*).
 */
#include <gnuradio/fft/window.h>
#include <gnuradio/io_signature.h>

#include "hip_sink_c_impl.h"

namespace gr {
namespace fosphor {

#ifdef FOSPHOR_AMD_HIP_CORE_ONLY
/* When base_sink_c_impl.cc (the GL implementation, which also defines this constructor at :36-44) is
 * left out of the build, the interface class still needs it. */
base_sink_c::base_sink_c(const char* name)
    : gr::sync_block(name, gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(0, 0, 0))
{
    message_port_register_out(pmt::mp("freq"));
}
#endif

hip_sink_c::sptr hip_sink_c::make() { return gnuradio::make_block_sptr<hip_sink_c_impl>(); }

hip_sink_c_impl::hip_sink_c_impl() : base_sink_c("hip_sink_c"), d_center(0.0), d_span(1.0)
{
    /* the GL sinks show the picture from the start; here "visible" only means that the worker
     * synchronises once per frame (fosphor_draw), which is what a polling front end wants */
    d_rt.set_visible(true);
}

hip_sink_c_impl::~hip_sink_c_impl() {}

void hip_sink_c_impl::execute_ui_action(enum ui_action_t action)
{
    /* same enumerator order as base_sink_c.h:35-48 */
    d_rt.execute_ui_action(static_cast<::fosphor_amd::sink_runtime::ui_action_t>(action));
}

void hip_sink_c_impl::execute_mouse_action(enum mouse_action_t action, int x, int y)
{
    /* base_sink_c_impl.cc:371-397: a click inside the spectrum pane publishes the frequency under the
     * pointer.  Without a window there is no pointer; a front end that has one calls fosphor_pos2freq()
     * on core() with its own fosphor_render and publishes on "freq" itself. */
    (void)action; (void)x; (void)y;
}

void hip_sink_c_impl::set_frequency_range(const double center, const double span)
{
    d_center = center; d_span = span;
    d_rt.set_frequency_range(center, span);
}

void hip_sink_c_impl::set_frequency_center(const double center)
{
    d_center = center;
    d_rt.set_frequency_center(center);
}

void hip_sink_c_impl::set_frequency_span(const double span)
{
    d_span = span;
    d_rt.set_frequency_span(span);
}

void hip_sink_c_impl::set_fft_window(const gr::fft::window::win_type win)
{
    /* base_sink_c_impl.cc:251-255: the taps come from GNU Radio, the core takes the 1024 floats */
    std::vector<float> taps = gr::fft::window::build(win, 1024, 6.76);
    d_rt.set_fft_window(taps.data());
}

int hip_sink_c_impl::work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star&)
{
    return d_rt.work(noutput_items, static_cast<const gr_complex*>(input_items[0]));
}

bool hip_sink_c_impl::start()
{
    const bool ok = base_sink_c::start();
    return d_rt.start() && ok;
}

bool hip_sink_c_impl::stop()
{
    const bool ok = d_rt.stop();
    return base_sink_c::stop() && ok;
}

} // namespace fosphor
} // namespace gr
