// Diagnostic: fill the vector register file of every SIMD with a pattern and exit, so that a kernel launched next which READS A
// REGISTER IT NEVER WROTE sees another value than it usually finds there.  (Round 4, looking for the cause of the round-3 stem
// weight-gradient irreproducibility next to a second process.)
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/experiments/vgpr_poison.hip -o tools/experiments/libvgpr_poison.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void poison_kernel(unsigned pattern, unsigned *sink) {
    const unsigned v = pattern ^ (threadIdx.x * 2654435761u);
    // registers v8 .. v207 written by name (clobber lists make the compiler allocate them)
    asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0" ::"v"(v) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
    asm volatile("v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0" ::"v"(v) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
    asm volatile("v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0" ::"v"(v) : "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    asm volatile("v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0" ::"v"(v) : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
    asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0" ::"v"(v) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    asm volatile("v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0" ::"v"(v) : "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    asm volatile("v_mov_b32 v56, %0\n v_mov_b32 v57, %0\n v_mov_b32 v58, %0\n v_mov_b32 v59, %0\n v_mov_b32 v60, %0\n v_mov_b32 v61, %0\n v_mov_b32 v62, %0\n v_mov_b32 v63, %0" ::"v"(v) : "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
    asm volatile("v_mov_b32 v64, %0\n v_mov_b32 v65, %0\n v_mov_b32 v66, %0\n v_mov_b32 v67, %0\n v_mov_b32 v68, %0\n v_mov_b32 v69, %0\n v_mov_b32 v70, %0\n v_mov_b32 v71, %0" ::"v"(v) : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    asm volatile("v_mov_b32 v72, %0\n v_mov_b32 v73, %0\n v_mov_b32 v74, %0\n v_mov_b32 v75, %0\n v_mov_b32 v76, %0\n v_mov_b32 v77, %0\n v_mov_b32 v78, %0\n v_mov_b32 v79, %0" ::"v"(v) : "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
    asm volatile("v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %0\n v_mov_b32 v83, %0\n v_mov_b32 v84, %0\n v_mov_b32 v85, %0\n v_mov_b32 v86, %0\n v_mov_b32 v87, %0" ::"v"(v) : "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    asm volatile("v_mov_b32 v88, %0\n v_mov_b32 v89, %0\n v_mov_b32 v90, %0\n v_mov_b32 v91, %0\n v_mov_b32 v92, %0\n v_mov_b32 v93, %0\n v_mov_b32 v94, %0\n v_mov_b32 v95, %0" ::"v"(v) : "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
    asm volatile("v_mov_b32 v96, %0\n v_mov_b32 v97, %0\n v_mov_b32 v98, %0\n v_mov_b32 v99, %0\n v_mov_b32 v100, %0\n v_mov_b32 v101, %0\n v_mov_b32 v102, %0\n v_mov_b32 v103, %0" ::"v"(v) : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103");
    asm volatile("v_mov_b32 v104, %0\n v_mov_b32 v105, %0\n v_mov_b32 v106, %0\n v_mov_b32 v107, %0\n v_mov_b32 v108, %0\n v_mov_b32 v109, %0\n v_mov_b32 v110, %0\n v_mov_b32 v111, %0" ::"v"(v) : "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
    asm volatile("v_mov_b32 v112, %0\n v_mov_b32 v113, %0\n v_mov_b32 v114, %0\n v_mov_b32 v115, %0\n v_mov_b32 v116, %0\n v_mov_b32 v117, %0\n v_mov_b32 v118, %0\n v_mov_b32 v119, %0" ::"v"(v) : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
    asm volatile("v_mov_b32 v120, %0\n v_mov_b32 v121, %0\n v_mov_b32 v122, %0\n v_mov_b32 v123, %0\n v_mov_b32 v124, %0\n v_mov_b32 v125, %0\n v_mov_b32 v126, %0\n v_mov_b32 v127, %0" ::"v"(v) : "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    asm volatile("v_mov_b32 v128, %0\n v_mov_b32 v129, %0\n v_mov_b32 v130, %0\n v_mov_b32 v131, %0\n v_mov_b32 v132, %0\n v_mov_b32 v133, %0\n v_mov_b32 v134, %0\n v_mov_b32 v135, %0" ::"v"(v) : "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135");
    asm volatile("v_mov_b32 v136, %0\n v_mov_b32 v137, %0\n v_mov_b32 v138, %0\n v_mov_b32 v139, %0\n v_mov_b32 v140, %0\n v_mov_b32 v141, %0\n v_mov_b32 v142, %0\n v_mov_b32 v143, %0" ::"v"(v) : "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
    asm volatile("v_mov_b32 v144, %0\n v_mov_b32 v145, %0\n v_mov_b32 v146, %0\n v_mov_b32 v147, %0\n v_mov_b32 v148, %0\n v_mov_b32 v149, %0\n v_mov_b32 v150, %0\n v_mov_b32 v151, %0" ::"v"(v) : "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151");
    asm volatile("v_mov_b32 v152, %0\n v_mov_b32 v153, %0\n v_mov_b32 v154, %0\n v_mov_b32 v155, %0\n v_mov_b32 v156, %0\n v_mov_b32 v157, %0\n v_mov_b32 v158, %0\n v_mov_b32 v159, %0" ::"v"(v) : "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159");
    asm volatile("v_mov_b32 v160, %0\n v_mov_b32 v161, %0\n v_mov_b32 v162, %0\n v_mov_b32 v163, %0\n v_mov_b32 v164, %0\n v_mov_b32 v165, %0\n v_mov_b32 v166, %0\n v_mov_b32 v167, %0" ::"v"(v) : "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167");
    asm volatile("v_mov_b32 v168, %0\n v_mov_b32 v169, %0\n v_mov_b32 v170, %0\n v_mov_b32 v171, %0\n v_mov_b32 v172, %0\n v_mov_b32 v173, %0\n v_mov_b32 v174, %0\n v_mov_b32 v175, %0" ::"v"(v) : "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175");
    asm volatile("v_mov_b32 v176, %0\n v_mov_b32 v177, %0\n v_mov_b32 v178, %0\n v_mov_b32 v179, %0\n v_mov_b32 v180, %0\n v_mov_b32 v181, %0\n v_mov_b32 v182, %0\n v_mov_b32 v183, %0" ::"v"(v) : "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183");
    asm volatile("v_mov_b32 v184, %0\n v_mov_b32 v185, %0\n v_mov_b32 v186, %0\n v_mov_b32 v187, %0\n v_mov_b32 v188, %0\n v_mov_b32 v189, %0\n v_mov_b32 v190, %0\n v_mov_b32 v191, %0" ::"v"(v) : "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191");
    asm volatile("v_mov_b32 v192, %0\n v_mov_b32 v193, %0\n v_mov_b32 v194, %0\n v_mov_b32 v195, %0\n v_mov_b32 v196, %0\n v_mov_b32 v197, %0\n v_mov_b32 v198, %0\n v_mov_b32 v199, %0" ::"v"(v) : "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199");
    asm volatile("v_mov_b32 v200, %0\n v_mov_b32 v201, %0\n v_mov_b32 v202, %0\n v_mov_b32 v203, %0\n v_mov_b32 v204, %0\n v_mov_b32 v205, %0\n v_mov_b32 v206, %0\n v_mov_b32 v207, %0" ::"v"(v) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207");
    if (v == 0x12345u && sink != nullptr) sink[0] = v;
}

extern "C" int lad_poison_vgprs(unsigned pattern, void *sink, void *stream) {
    hipLaunchKernelGGL(poison_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, pattern, (unsigned *)sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
