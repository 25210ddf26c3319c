// SceneMarshal.h — host scene description -> flat, pointer-free rtw_scene blob (include/rtw.h).
// This is the counterpart of the reference's marshalling code:
//   Director::createSBT        (Director.cpp:628-885)  geometry records
//   Director::initLaunchParams (Director.cpp:483-553)  camera, materials+textures, lights, pdf tree
// including what that code does NOT copy: the lens radius stays 0 (Director.cpp:494-496 vs
// sysparameter.h:50) and the shutter interval is hard-wired to [0,1] (Director.cpp:521-522).
#pragma once
#include <cstring>
#include <vector>

#include "ioScene.h"

namespace rtwhost {

inline size_t align16(size_t v) { return (v + 15u) & ~size_t(15); }

inline std::vector<uint8_t> marshalScene(const ioScene& sc) {
    std::vector<rtw_prim> prims;
    std::vector<rtw_xform> xforms;
    std::vector<rtw_material> mats;
    TexEmit emit;  // texture records + texture data section, in material-list order (Director.cpp:503-513)

    rtw_xform ident{};
    ident.m[0] = ident.m[5] = ident.m[10] = 1.f;
    ident.inv[0] = ident.inv[5] = ident.inv[10] = 1.f;
    xforms.push_back(ident);

    // materials: one record per instance id, each with its own texture record (TexArray, Director.cpp:503-513)
    for (size_t i = 0; i < sc.materialList.size(); i++) {
        rtw_material m{};
        m.texture = -1;
        const ioTexture* t = sc.materialList[i]->assignTo(m);
        if (t) m.texture = t->emit(emit);
        mats.push_back(m);
    }

    for (size_t i = 0; i < sc.geoInstList.size(); i++) {
        const ioGeometryInstance& gi = sc.geoInstList[i];
        rtw_prim p = sc.geometryList[gi.sbtOffset]->record();
        p.material = static_cast<int32_t>(gi.instanceId);
        p.xform = 0;
        if (!gi.identity) {
            int found = -1;
            for (size_t k = 1; k < xforms.size(); k++)
                if (!memcmp(xforms[k].m, gi.transform.data(), sizeof(float) * 12)) { found = static_cast<int>(k); break; }
            if (found < 0) {
                rtw_xform x{};
                memcpy(x.m, gi.transform.data(), sizeof(float) * 12);
                memcpy(x.inv, gi.inv.data(), sizeof(float) * 12);
                found = static_cast<int>(xforms.size());
                xforms.push_back(x);
            }
            p.xform = found;
        }
        prims.push_back(p);
    }

    rtw_scene_header h{};
    h.magic = RTW_SCENE_MAGIC;
    h.version = RTW_SCENE_VERSION;
    h.n_prims = static_cast<uint32_t>(prims.size());
    h.n_xforms = static_cast<uint32_t>(xforms.size());
    h.n_materials = static_cast<uint32_t>(mats.size());
    const std::vector<rtw_texture>& texs = emit.texs;
    h.n_textures = static_cast<uint32_t>(texs.size());
    h.n_lights = static_cast<uint32_t>(sc.m_lightDefinitions.size());
    h.sky_light = sc.m_lightDefinitions.empty() ? 1 : 0;  // Director.cpp:523

    Float3 o, u, v, w, llc, hor, ver;
    sc.camera->getfrustum(o, u, v, w, llc, hor, ver);
    auto put = [](float* d, const Float3& s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    put(h.camera.origin, o); put(h.camera.u, u); put(h.camera.v, v); put(h.camera.w, w);
    put(h.camera.lower_left, llc); put(h.camera.horizontal, hor); put(h.camera.vertical, ver);
    h.camera.lens_radius = 0.f;
    h.camera_type = sc.camera->cameraType();
    h.camera.time0 = 0.f;
    h.camera.time1 = 1.f;

    h.pdf.gen = sc.MCpdf.pdfGenIdx;
    h.pdf.p0_gen = sc.MCpdf.p0GenIdx;
    h.pdf.p1_gen = sc.MCpdf.p1GenIdx;
    h.pdf.flip = sc.MCpdf.flip;
    h.pdf.bias = sc.MCpdf.bias;
    memcpy(h.pdf.rect, sc.MCpdf.pdfrect, sizeof(float) * 5);

    size_t off = align16(sizeof(rtw_scene_header));
    h.off_prims = static_cast<uint32_t>(off); off = align16(off + prims.size() * sizeof(rtw_prim));
    h.off_xforms = static_cast<uint32_t>(off); off = align16(off + xforms.size() * sizeof(rtw_xform));
    h.off_materials = static_cast<uint32_t>(off); off = align16(off + mats.size() * sizeof(rtw_material));
    h.off_textures = static_cast<uint32_t>(off); off = align16(off + texs.size() * sizeof(rtw_texture));
    h.off_lights = static_cast<uint32_t>(off); off = align16(off + sc.m_lightDefinitions.size() * sizeof(rtw_light));
    if (!emit.data.empty()) {  // texture data section (noise tables, image texels); absent for constant-only scenes
        h.off_texdata = static_cast<uint32_t>(off);
        h.texdata_bytes = static_cast<uint32_t>(emit.data.size() * sizeof(uint32_t));
        off = align16(off + h.texdata_bytes);
    }
    h.total_bytes = static_cast<uint32_t>(off);

    std::vector<uint8_t> blob(off, 0);
    memcpy(blob.data(), &h, sizeof h);
    if (!prims.empty()) memcpy(blob.data() + h.off_prims, prims.data(), prims.size() * sizeof(rtw_prim));
    memcpy(blob.data() + h.off_xforms, xforms.data(), xforms.size() * sizeof(rtw_xform));
    if (!mats.empty()) memcpy(blob.data() + h.off_materials, mats.data(), mats.size() * sizeof(rtw_material));
    if (!texs.empty()) memcpy(blob.data() + h.off_textures, texs.data(), texs.size() * sizeof(rtw_texture));
    if (!sc.m_lightDefinitions.empty())
        memcpy(blob.data() + h.off_lights, sc.m_lightDefinitions.data(), sc.m_lightDefinitions.size() * sizeof(rtw_light));
    if (!emit.data.empty()) memcpy(blob.data() + h.off_texdata, emit.data.data(), h.texdata_bytes);
    return blob;
}

}  // namespace rtwhost
