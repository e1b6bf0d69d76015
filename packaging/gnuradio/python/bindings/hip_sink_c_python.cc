/* pybind11 binding of gr::fosphor::hip_sink_c.  NOT COMPILED in the build container (no GNU Radio). */
#include <pybind11/complex.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

namespace py = pybind11;

#include <gnuradio/fosphor/hip_sink_c.h>

void bind_hip_sink_c(py::module& m)
{
    using hip_sink_c = gr::fosphor::hip_sink_c;

    py::class_<hip_sink_c, gr::fosphor::base_sink_c, gr::sync_block, gr::block, gr::basic_block,
               std::shared_ptr<hip_sink_c>>(m, "hip_sink_c")
        .def(py::init(&hip_sink_c::make))
        .def("frames", &hip_sink_c::frames)
        .def("samples_processed", &hip_sink_c::samples_processed)
        /* the core as an integer handle for ctypes users of libfosphor_amd.so (gr-fosphor_amd/_lib.py) */
        .def("core_handle", [](hip_sink_c& s) { return reinterpret_cast<uintptr_t>(s.core()); });
}
