/* gr-fosphor: headless sink on the HIP compute core.  NOT COMPILED in the build container (no GNU Radio). */
#pragma once

#include <gnuradio/fosphor/api.h>
#include <gnuradio/fosphor/base_sink_c.h>

struct fosphor;

namespace gr {
namespace fosphor {

/*!
 * \brief fosphor sink without a window: the spectrum state lives in GPU memory
 * \ingroup fosphor
 *
 * Same stream input, "freq" message port, UI actions and setters as the GLFW / Qt sinks
 * (base_sink_c.h:35-59); instead of drawing it exposes the compute core, whose plain device
 * buffers (fosphor_amd_get_buffers) or coloured images (fosphor_amd_colorize) a front end maps.
 */
class GR_FOSPHOR_API hip_sink_c : virtual public base_sink_c
{
public:
    typedef std::shared_ptr<hip_sink_c> sptr;
    static sptr make();

    /*! the core, for fosphor_amd_get_buffers / fosphor_amd_colorize / fosphor_amd_freq_labels */
    virtual struct ::fosphor* core() = 0;
    /*! frames rendered and samples consumed so far */
    virtual uint64_t frames() const = 0;
    virtual uint64_t samples_processed() const = 0;
};

} // namespace fosphor
} // namespace gr
