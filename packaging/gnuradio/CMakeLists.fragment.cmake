# --- append to lib/CMakeLists.txt of gr-fosphor (not compiled in the build container: no GNU Radio there) ---
option(ENABLE_AMD_HIP_CORE "Compute core: libfosphor_amd (HIP, MI355X) instead of the OpenCL kernels" OFF)

if(ENABLE_AMD_HIP_CORE)
  set(FOSPHOR_AMD_ROOT "" CACHE PATH "checkout of the HIP compute core (contains include/ and gr-fosphor_amd/)")
  find_library(FOSPHOR_AMD_LIB fosphor_amd HINTS ${FOSPHOR_AMD_ROOT}/gr-fosphor_amd REQUIRED)

  # the OpenCL half and its kernel resources leave the module; the GL front end stays
  list(REMOVE_ITEM fosphor_sources fosphor/fosphor.c fosphor/cl.c fosphor/cl_compat.c)
  list(APPEND      fosphor_sources hip_sink_c_impl.cc)

  target_include_directories(gnuradio-fosphor PRIVATE ${FOSPHOR_AMD_ROOT}/include)
  target_compile_definitions(gnuradio-fosphor PRIVATE FOSPHOR_AMD_HIP_CORE=1)
  target_link_libraries(gnuradio-fosphor ${FOSPHOR_AMD_LIB})
  install(FILES ${CMAKE_SOURCE_DIR}/include/gnuradio/fosphor/hip_sink_c.h DESTINATION include/gnuradio/fosphor)
endif()
