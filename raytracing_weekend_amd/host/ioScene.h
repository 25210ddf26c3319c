// ioScene.h — the hard-coded scenes, as in the reference's scene/ioScene.h:
//   0 CornellBox          (ioScene.h:491-627)   metric scene
//   1 MovingSpheres       (ioScene.h:180-309)   "In One Weekend" final scene, 70% moving spheres
//   2 InOneWeekendLight   (ioScene.h:313-489)   static spheres, Perlin ground, earth-map sphere, one rectangle light
//   3 VolumesCornellBox   (ioScene.h:630-788)   Cornell box with two participating media
//   4 TheNextWeekFinal    (ioScene.h:791-982)   400 ground boxes, 1000-sphere cluster under a transform, two media,
//                                               noise / image textures, a moving sphere (3410 primitives)
// Scenes 2 and 4 load assets/earthmap.jpg where one is installed (the reference's asset, decoded by JpegDecode.h), else the
// synthetic assets/earthmap.ppm of this tree (see earthMapPath).
// Primitive i, material i and instance i line up, as the reference relies on
// (geometryList.size()==materialList.size(), ioScene.h:262,426,581).
#pragma once
#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "ioCamera.h"
#include "ioGeometry.h"

namespace rtwhost {

// lib/random.cuh:22-38 — the scene layouts are drawn with the reference's xorshift32/randf.
inline uint32_t xorshift32(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}
inline float randf(uint32_t& s) {
    float r = static_cast<float>(xorshift32(s)) / 4294967296.0f;
    return r != 1.0f ? r : static_cast<float>(0x3F7FFFFF);
}

// sysparameter.h:18-30 / ioScene.h:35-46, flattened: generate ids of the pdf tree
struct pdfCallfun_host {
    int pdfGenIdx = RTW_PDF_COSINE;
    int p0GenIdx = -1;
    int p1GenIdx = -1;
    float pdfrect[5] = {0, 0, 0, 0, 0};
    int flip = 0;
    float bias = 0.f;
};

// Where an asset of the reference's assets/ directory is looked for: $RTW_ASSET_DIR, ./assets (the reference's
// relative path), then assets/ next to the binary this code is linked into (librtw_host.so or rtw_render).
inline std::string assetPath(const std::string& name) {
    std::vector<std::string> dirs;
    if (const char* e = getenv("RTW_ASSET_DIR")) dirs.push_back(e);
    dirs.push_back("assets");
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&xorshift32), &info) && info.dli_fname) {
        std::string self(info.dli_fname);
        const size_t slash = self.find_last_of('/');
        dirs.push_back((slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/assets");
    }
    for (const std::string& d : dirs) {
        const std::string p = d + "/" + name;
        if (std::ifstream(p).good()) return p;
    }
    return dirs.back() + "/" + name;  // reported in the error message of the loader
}
// the reference's earth map (assets/earthmap.jpg, ioScene.h:438,911) where it is installed, else the synthetic PPM this tree ships
inline std::string earthMapPath() {
    const std::string jpg = assetPath("earthmap.jpg");
    if (std::ifstream(jpg).good()) return jpg;
    return assetPath("earthmap.ppm");
}

class ioScene {
public:
    // returns non-zero for an unknown scene, like ioScene::init (ioScene.h:52-101), and when an asset is missing
    int init(int Nx, int Ny, int Ns, int maxRayDepth, int Nscene) {
        m_Nx = Nx; m_Ny = Ny; m_numSamples = Ns; m_maxRayDepth = maxRayDepth;
        destroy();
        if (getScenePdf(Nscene)) return 1;
        try {
            switch (Nscene) {
            case 0: CornellBox(); break;
            case 1: MovingSpheres(); break;
            case 2: InOneWeekendLight(); break;
            case 3: VolumesCornellBox(); break;
            case 4: TheNextWeekFinal(); break;
            default:
                std::cerr << "ERROR: Scene " << Nscene << " unknown." << std::endl;
                return 1;
            }
        } catch (const std::exception& ex) {  // the reference dereferences stbi_load's null result instead
            std::cerr << "ERROR: " << ex.what() << std::endl;
            destroy();
            return 1;
        }
        return 0;
    }

    void destroy() {
        geometryList.clear(); materialList.clear(); geoInstList.clear(); m_lightDefinitions.clear();
        ownedTextures.clear(); ownedMaterials.clear(); camera.reset();
    }

    std::string getDescription() const { return sceneDescription; }

    std::vector<std::unique_ptr<ioGeometry>> geometryList;
    std::vector<const ioMaterial*> materialList;
    std::vector<ioGeometryInstance> geoInstList;
    std::vector<rtw_light> m_lightDefinitions;
    pdfCallfun_host MCpdf;
    std::unique_ptr<ioCamera> camera;
    // swap the scene's perspective camera for one of the reference's two unused kinds at the same position
    // and frame. The orthographic window is the perspective image plane (its extent at the focus distance), and - because
    // scene/camera.cuh:52 adds the camera origin to a corner that already contains it - the frame is re-centred on the origin
    // (position 0) so that the window looks at the scene; kind: rtw_camera_type.
    void setCameraKind(int kind) {
        if (!camera || kind == RTW_CAM_PERSPECTIVE) return;
        Float3 o, u, v, w, llc, hor, ver;
        camera->getfrustum(o, u, v, w, llc, hor, ver);
        const Float3 at = o - w, up = v;
        if (kind == RTW_CAM_ENVIRONMENT) {
            camera.reset(new ioEnvironmentCamera(o.x, o.y, o.z, at.x, at.y, at.z, up.x, up.y, up.z, camera->m_time0, camera->m_time1));
        } else if (kind == RTW_CAM_ORTHOGRAPHIC) {
            std::unique_ptr<ioOrthographicCamera> oc(new ioOrthographicCamera(o.x, o.y, o.z, at.x, at.y, at.z, up.x, up.y, up.z, length(ver), length(hor),
                                                                              camera->m_time0, camera->m_time1));
            // corner relative to the position; the device adds the position back (camera.cuh:52)
            oc->m_lowerLeftCorner = oc->m_lowerLeftCorner - oc->m_origin;
            camera = std::move(oc);
        }
    }

private:
    // ioScene.h:103-148
    int getScenePdf(int Nscene) {
        MCpdf = pdfCallfun_host();
        switch (Nscene) {
        case 0:
            MCpdf.pdfGenIdx = RTW_PDF_MIXTURE; MCpdf.p0GenIdx = RTW_PDF_COSINE; MCpdf.p1GenIdx = RTW_PDF_RECT_Y;
            setRect(213.f, 343.f, 227.f, 332.f, 554.9f);
            return 0;
        case 1:
            MCpdf.pdfGenIdx = RTW_PDF_COSINE;
            return 0;
        case 2:  // note the rectangle the pdf samples (y 2.3..6) is not the light's (y 1..3): ioScene.h:121-127
            MCpdf.pdfGenIdx = RTW_PDF_MIXTURE; MCpdf.p0GenIdx = RTW_PDF_COSINE; MCpdf.p1GenIdx = RTW_PDF_RECT_Z;
            setRect(3.f, 5.f, 2.3f, 3.f + 3.f, -2.0f);
            return 0;
        case 3:
            MCpdf.pdfGenIdx = RTW_PDF_MIXTURE; MCpdf.p0GenIdx = RTW_PDF_COSINE; MCpdf.p1GenIdx = RTW_PDF_RECT_Y;
            setRect(213.f, 343.f, 227.f, 332.f, 554.f);
            return 0;
        case 4:
            MCpdf.pdfGenIdx = RTW_PDF_MIXTURE; MCpdf.p0GenIdx = RTW_PDF_COSINE; MCpdf.p1GenIdx = RTW_PDF_RECT_Y;
            setRect(123.f, 423.f, 147.f, 412.f, 554.f);
            return 0;
        default:
            std::cerr << "ERROR: Scene " << Nscene << " unknown." << std::endl;
            return 1;
        }
    }
    void setRect(float a0, float a1, float b0, float b1, float k) {
        MCpdf.pdfrect[0] = a0; MCpdf.pdfrect[1] = a1; MCpdf.pdfrect[2] = b0; MCpdf.pdfrect[3] = b1; MCpdf.pdfrect[4] = k;
    }

    const ioTexture* tex(ioTexture* t) { ownedTextures.emplace_back(t); return t; }
    const ioMaterial* mat(ioMaterial* m) { ownedMaterials.emplace_back(m); return m; }

    void identityInstances() {
        geoInstList.resize(geometryList.size());
        for (size_t i = 0; i < geoInstList.size(); i++) geoInstList[i].init(static_cast<unsigned>(i), static_cast<unsigned>(i));
    }

    // ---------------------------------------------------------------- scene 0
    void CornellBox() {
        sceneDescription = "Cornell box";
        const ioMaterial* wallRed = mat(new ioLambertianMaterial(tex(new ioConstantTexture(make_float3(0.65f, 0.05f, 0.05f)))));
        const ioMaterial* wallGreen = mat(new ioLambertianMaterial(tex(new ioConstantTexture(make_float3(0.12f, 0.45f, 0.15f)))));
        const ioMaterial* wallWhite = mat(new ioLambertianMaterial(tex(new ioConstantTexture(make_float3(0.73f, 0.73f, 0.73f)))));
        const ioMaterial* aluminum = mat(new ioMetalMaterial(tex(new ioConstantTexture(make_float3(0.91f, 0.92f, 0.92f))), 0.018f));
        const ioTexture* light15 = tex(new ioConstantTexture(make_float3(15.f, 15.f, 15.f)));

        geometryList.emplace_back(new ioSphere(190.f, 90.f, 190.f, 90.f));  // medium glass sphere
        materialList.push_back(mat(new ioDielectricMaterial(1.5f)));

        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 555.f, true, X_AXIS));   // left wall
        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 0.f, false, X_AXIS));    // right wall
        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 555.f, true, Y_AXIS));   // roof
        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 0.f, false, Y_AXIS));    // floor
        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 555.f, true, Z_AXIS));   // back wall
        geometryList.emplace_back(new ioAARect(213.f, 343.f, 227.f, 332.f, 554.9f, true, Y_AXIS));  // light
        materialList.push_back(wallGreen);
        materialList.push_back(wallRed);
        materialList.push_back(wallWhite);
        materialList.push_back(wallWhite);
        materialList.push_back(wallWhite);
        materialList.push_back(mat(new ioDiffuseLightMaterial(light15)));

        // aluminium box: six rects under T(265,0,295) * R_y(15 deg)   (ioScene.h:537-548)
        Float3 b1size = make_float3(165.f, 330.f, 165.f);
        Float3 b1tr = make_float3(265.f, 0.f, 295.f);
        ioGeometryGroup::createBox(make_float3(0.f), b1size, geometryList);
        for (int i = 0; i < 6; i++) materialList.push_back(aluminum);
        Mat4 transf = ioTransform::translate(b1tr);
        transf *= ioTransform::rotateY(15.f);

        identityInstances();
        for (size_t i = 7; i < 13; i++) geoInstList[i].setTransform(transf);

        // ioScene.h:605-612
        rtw_light light{};
        Float3 vecU = make_float3(343.f - 213.f, 0.f, 0.f), vecV = make_float3(0.f, 0.f, 332.f - 227.f);
        Float3 c = cross(vecU, vecV), n = normalize(c);
        light.emission[0] = light.emission[1] = light.emission[2] = 15.f;
        light.vec_u[0] = vecU.x; light.vec_u[1] = vecU.y; light.vec_u[2] = vecU.z;
        light.vec_v[0] = vecV.x; light.vec_v[1] = vecV.y; light.vec_v[2] = vecV.z;
        light.position[0] = 213.f; light.position[1] = 554.f; light.position[2] = 227.f;
        light.area = length(c);
        light.normal[0] = n.x; light.normal[1] = n.y; light.normal[2] = n.z;
        m_lightDefinitions.push_back(light);

        camera.reset(new ioPerspectiveCamera(278.f, 278.f, -800.f, 278.f, 278.f, 0.f, 0.0f, 1.0f, 0.0f, 40.0f,
                                             float(m_Nx) / float(m_Ny), /*aperture*/ 1.f, /*focus_distance*/ 10.f, 0.f, 1.f));
    }

    // The 22 x 22 grid of small spheres of scenes 1 and 2 (ioScene.h:201-253, 373-422): one material draw, a jittered
    // position, a keep-out zone around the medium spheres; 70 % diffuse (bobbing upwards by 0.18 in scene 1), 15 % metal,
    // 8 % glass, 7 % hollow glass (two concentric spheres). Where the reference draws several randf() inside one
    // argument list the order is unspecified C++; it only builds with MSVC, which evaluates arguments right
    // to left, so colours are drawn blue, green, red and a metal's fuzz before its colour.
    void scatterSmallSpheres(uint32_t seed, bool moving) {
        const float r = 0.2f;
        for (int a = -11; a < 11; a++) {
            for (int b = -11; b < 11; b++) {
                const float chooseMat = randf(seed);
                const float x = a + 0.8f * randf(seed);
                const float y = 0.2f;
                const float z = b + 0.9f * randf(seed);
                const float z_squared = z * z;
                const float dist = sqrtf((x - 4.0f) * (x - 4.0f) + z_squared);
                if (!((dist > 0.9f) || ((z_squared > 0.7f) && ((x * x - 16.0f) > -2.f)))) continue;
                if (chooseMat < 0.70f) {
                    if (moving) geometryList.emplace_back(new ioMovingSphere(x, y, z, x, y + 0.18f, z, r, 0.f, 1.f));
                    else geometryList.emplace_back(new ioSphere(x, y, z, r));
                    const float cb = randf(seed), cg = randf(seed), cr = randf(seed);
                    materialList.push_back(mat(new ioLambertianMaterial(tex(new ioConstantTexture(make_float3(cr, cg, cb))))));
                } else if (chooseMat < 0.85f) {
                    geometryList.emplace_back(new ioSphere(x, y, z, r));
                    const float fuzz = 0.5f * randf(seed);
                    const float cb = 0.5f * (1.0f - randf(seed));
                    const float cg = 0.5f * (1.0f - randf(seed));
                    const float cr = 0.5f * (1.0f - randf(seed));
                    materialList.push_back(mat(new ioMetalMaterial(tex(new ioConstantTexture(make_float3(cr, cg, cb))), fuzz)));
                } else {
                    geometryList.emplace_back(new ioSphere(x, y, z, r));
                    materialList.push_back(mat(new ioDielectricMaterial(1.5f)));
                    if (!(chooseMat < 0.93f)) {  // hollow: a second, slightly smaller glass sphere
                        geometryList.emplace_back(new ioSphere(x, y, z, (r - 0.007f)));
                        materialList.push_back(mat(new ioDielectricMaterial(1.5f)));
                    }
                }
            }
        }
    }

    // ---------------------------------------------------------------- scene 1
    void MovingSpheres() {
        sceneDescription = "InOneWeekend final scene with moving spheres";
        const ioTexture* fiftyPercentGrey = tex(new ioConstantTexture(make_float3(0.5f, 0.5f, 0.5f)));
        const ioTexture* fiftyPercentReddishGrey = tex(new ioConstantTexture(make_float3(0.7f, 0.6f, 0.5f)));
        const ioTexture* reddish = tex(new ioConstantTexture(make_float3(0.4f, 0.2f, 0.1f)));

        geometryList.emplace_back(new ioSphere(0.0f, -1000.0f, 0.0f, 1000.0f));  // ground
        materialList.push_back(mat(new ioLambertianMaterial(fiftyPercentGrey)));
        geometryList.emplace_back(new ioSphere(0.0f, 1.0f, 0.0f, 1.0f));
        geometryList.emplace_back(new ioSphere(-4.0f, 1.0f, 0.0f, 1.0f));
        geometryList.emplace_back(new ioSphere(4.0f, 1.0f, 0.0f, 1.0f));
        materialList.push_back(mat(new ioDielectricMaterial(1.5f)));
        materialList.push_back(mat(new ioLambertianMaterial(reddish)));
        materialList.push_back(mat(new ioMetalMaterial(fiftyPercentReddishGrey, 0.1f)));

        scatterSmallSpheres(0x314759, /*moving*/ true);  // ioScene.h:201-253
        identityInstances();
        camera.reset(new ioPerspectiveCamera(13.0f, 2.0f, 3.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 20.0f,
                                             float(m_Nx) / float(m_Ny), /*aperture*/ 0.1f, /*focus_distance*/ 10.f, 0.f, 1.f));
        // no light definition: the sky lights the scene (Director.cpp:523-524)
    }

    void rectLight(const Float3& pos, const Float3& U, const Float3& V, const Float3& emission) {
        rtw_light light{};
        light.emission[0] = emission.x; light.emission[1] = emission.y; light.emission[2] = emission.z;
        light.vec_u[0] = U.x; light.vec_u[1] = U.y; light.vec_u[2] = U.z;
        light.vec_v[0] = V.x; light.vec_v[1] = V.y; light.vec_v[2] = V.z;
        light.position[0] = pos.x; light.position[1] = pos.y; light.position[2] = pos.z;
        Float3 c = cross(U, V);
        Float3 n = normalize(c);
        light.area = length(c);
        light.normal[0] = n.x; light.normal[1] = n.y; light.normal[2] = n.z;
        m_lightDefinitions.push_back(light);
    }

    // ---------------------------------------------------------------- scene 2
    void InOneWeekendLight() {
        sceneDescription = "IOW Scene with a light box";
        const ioTexture* constantGrey = tex(new ioConstantTexture(make_float3(0.7f, 0.7f, 0.7f)));
        const ioTexture* noise1 = tex(new ioNoiseTexture(1.f));
        const ioTexture* earthGlobeImage = tex(new ioImageTexture(earthMapPath()));
        const ioTexture* light16 = tex(new ioConstantTexture(make_float3(16.f, 16.f, 16.f)));

        geometryList.emplace_back(new ioSphere(0.0f, -1000.0f, 0.0f, 1000.0f));  // big sphere, Perlin ground
        materialList.push_back(mat(new ioLambertianMaterial(noise1)));
        geometryList.emplace_back(new ioSphere(-4.0f, 1.0f, 0.0f, 1.0f));
        geometryList.emplace_back(new ioSphere(0.0f, 1.0f, 0.0f, 1.0f));
        geometryList.emplace_back(new ioSphere(4.0f, 1.0f, 0.0f, 1.0f));
        materialList.push_back(mat(new ioMetalMaterial(constantGrey, 0.4f)));
        materialList.push_back(mat(new ioLambertianMaterial(earthGlobeImage)));
        materialList.push_back(mat(new ioDielectricMaterial(1.5f)));
        geometryList.emplace_back(new ioAARect(3.f, 5.f, 1.f, 3.f, -2.0f, false, Z_AXIS));
        materialList.push_back(mat(new ioDiffuseLightMaterial(light16)));
        rectLight(make_float3(3.f, 1.f, -2.f), make_float3(5.f - 3.f, 0.f, 0.f), make_float3(0.f, 3.f - 1.f, 0.f), make_float3(16.f, 16.f, 16.f));

        scatterSmallSpheres(0x6314759, /*moving*/ false);  // ioScene.h:373-422
        identityInstances();
        camera.reset(new ioPerspectiveCamera(13.0f, 2.0f, 3.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 20.0f,
                                             float(m_Nx) / float(m_Ny), /*aperture*/ 0.08f, /*focus_distance*/ 10.f));
    }

    // ---------------------------------------------------------------- scene 4
    void TheNextWeekFinal() {
        sceneDescription = "The Next Week final scene";
        const ioTexture* brown = tex(new ioConstantTexture(make_float3(0.7f, 0.3f, 0.1f)));
        const ioTexture* groundGreenish = tex(new ioConstantTexture(make_float3(0.48f, 0.83f, 0.53f)));
        const ioTexture* metal1 = tex(new ioConstantTexture(make_float3(0.8f, 0.8f, 0.9f)));
        const ioTexture* noisep1 = tex(new ioNoiseTexture(0.1f));
        const ioTexture* earthGlobeImage = tex(new ioImageTexture(earthMapPath()));
        const ioTexture* light7 = tex(new ioConstantTexture(make_float3(7.f, 7.f, 7.f)));
        uint32_t seed = 0x6314759;
        const ioMaterial* glassyBlueFog = mat(new ioIsotropicMaterial(tex(new ioConstantTexture(make_float3(0.2f, 0.4f, 0.9f)))));
        const ioMaterial* ambientFog = mat(new ioIsotropicMaterial(tex(new ioConstantTexture(make_float3(0.95f)))));
        const ioMaterial* ground = mat(new ioLambertianMaterial(groundGreenish));

        // instance id = material index, one geometry per instance; the order below is the order of the reference's
        // instance list (geoInstList, then the two media, then the sphere cluster: ioScene.h:947-955)
        auto instance = [&](const ioMaterial* m) -> ioGeometryInstance& {
            materialList.push_back(m);
            geoInstList.emplace_back();
            geoInstList.back().init(static_cast<unsigned>(materialList.size() - 1), static_cast<unsigned>(geometryList.size() - 1));
            return geoInstList.back();
        };
        geometryList.emplace_back(new ioAARect(123.f, 423.f, 147.f, 412.f, 554.f, true, Y_AXIS));  // light
        instance(mat(new ioDiffuseLightMaterial(light7)));
        rectLight(make_float3(123.f, 554.f, 147.f), make_float3(423.f - 123.f, 0.f, 0.f), make_float3(0.f, 0.f, 412.f - 147.f), make_float3(7.f, 7.f, 7.f));
        Float3 center = make_float3(400.f, 400.f, 200.f);
        Float3 center1tr = center + make_float3(30.f, 0.f, 0.f);
        geometryList.emplace_back(new ioSphere(260.f, 150.f, 45.f, 50.f));  // glass sphere
        instance(mat(new ioDielectricMaterial(1.5f)));
        geometryList.emplace_back(new ioSphere(0.f, 150.f, 145.f, 50.f));   // metal sphere
        instance(mat(new ioMetalMaterial(metal1, 0.2f)));
        Float3 centerGlassy = make_float3(360.f, 150.f, 45.f);
        geometryList.emplace_back(new ioSphere(centerGlassy.x, centerGlassy.y, centerGlassy.z, 70.f));  // blue glassy sphere (holds a medium)
        instance(mat(new ioDielectricMaterial(1.5f)));
        geometryList.emplace_back(new ioSphere(0.f, 0.f, 0.f, 5000.f));     // room boundary
        instance(mat(new ioDielectricMaterial(1.5f)));
        geometryList.emplace_back(new ioSphere(400.f, 200.f, 400.f, 100.0f));  // earth globe
        instance(mat(new ioLambertianMaterial(earthGlobeImage)));
        geometryList.emplace_back(new ioSphere(220.f, 280.f, 300.f, 80.f));    // marble
        instance(mat(new ioLambertianMaterial(noisep1)));
        geometryList.emplace_back(new ioMovingSphere(center.x, center.y, center.z, center1tr.x, center1tr.y, center1tr.z, 50.f, 0.f, 1.f));
        instance(mat(new ioLambertianMaterial(brown)));

        // ground: 20 x 20 boxes of random height, six rectangles each (ioScene.h:886-918)
        for (int i = 0; i < 20; i++) {
            for (int j = 0; j < 20; j++) {
                float w = 100.f;
                float x0 = -1000 + i * w;
                float z0 = -1000 + j * w;
                float y0 = 0.f;
                float x1 = x0 + w;
                float y1 = 100 * (randf(seed) + 0.01f);
                float z1 = z0 + w;
                std::vector<std::unique_ptr<ioGeometry>> box;
                ioGeometryGroup::createBox(make_float3(x0, y0, z0), make_float3(x1, y1, z1), box);
                for (auto& r : box) {
                    geometryList.emplace_back(std::move(r));
                    instance(ground);
                }
            }
        }
        // the two media (ioScene.h:920-929)
        geometryList.emplace_back(new ioVolumeSphere(centerGlassy.x, centerGlassy.y, centerGlassy.z, 70.f, 0.2f));
        instance(glassyBlueFog);
        geometryList.emplace_back(new ioVolumeSphere(0.f, 0.f, 0.f, 500.f, 8e-5f));
        instance(ambientFog);
        // 1000 small spheres in a 165-cube, moved as a block: T(-100,270,395) * R_y(20 deg) (ioScene.h:931-945)
        const ioMaterial* white = mat(new ioLambertianMaterial(tex(new ioConstantTexture(make_float3(0.93f)))));
        Mat4 transmat = ioTransform::translate(make_float3(-100.f, 270.f, 395.f));
        transmat *= ioTransform::rotateY(20.f);
        for (int j = 0; j < 1000; j++) {
            float cz = 165 * randf(seed), cy = 165 * randf(seed), cx = 165 * randf(seed);  // right to left (Q6)
            geometryList.emplace_back(new ioSphere(cx, cy, cz, 10.f));
            instance(white).setTransform(transmat);
        }
        camera.reset(new ioPerspectiveCamera(478.f, 278.f, -600.f, 278.f, 278.f, 0.f, 0.f, 1.f, 0.f, 40.0f,
                                             float(m_Nx) / float(m_Ny), /*aperture*/ 0.1f, /*focus_distance*/ 10.f, 0.f, 1.f));
    }

    // ---------------------------------------------------------------- scene 3
    void VolumesCornellBox() {
        sceneDescription = "Cornell box with volumes (participating media)";
        const ioMaterial* wallRed = mat(new ioLambertianMaterial(tex(new ioConstantTexture(make_float3(0.65f, 0.05f, 0.05f)))));
        const ioMaterial* wallGreen = mat(new ioLambertianMaterial(tex(new ioConstantTexture(make_float3(0.12f, 0.45f, 0.15f)))));
        const ioMaterial* wallWhite = mat(new ioLambertianMaterial(tex(new ioConstantTexture(make_float3(0.73f, 0.73f, 0.73f)))));
        const ioTexture* light15 = tex(new ioConstantTexture(make_float3(15.f, 15.f, 15.f)));
        const ioMaterial* blackFog = mat(new ioIsotropicMaterial(tex(new ioConstantTexture(make_float3(0.f)))));
        const ioMaterial* whiteFog = mat(new ioIsotropicMaterial(tex(new ioConstantTexture(make_float3(1.f)))));

        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 555.f, true, X_AXIS));
        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 0.f, false, X_AXIS));
        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 555.f, true, Y_AXIS));
        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 0.f, false, Y_AXIS));
        geometryList.emplace_back(new ioAARect(0.f, 555.f, 0.f, 555.f, 555.f, true, Z_AXIS));
        geometryList.emplace_back(new ioAARect(213.f, 343.f, 227.f, 332.f, 554.f, true, Y_AXIS));  // light
        materialList.push_back(wallGreen);
        materialList.push_back(wallRed);
        materialList.push_back(wallWhite);
        materialList.push_back(wallWhite);
        materialList.push_back(wallWhite);
        materialList.push_back(mat(new ioDiffuseLightMaterial(light15)));

        // black fog box, T(b1tr) * R_z(-12.5 deg) * R_y(15 deg)   (ioScene.h:694-715)
        const float z1Theta = -12.5f * (kPiF / 180.f);
        Float3 b1size = make_float3(165.f, 330.f, 165.f);
        Float3 b1tr = make_float3(265.f, fabsf(sinf(z1Theta)) * b1size.x + 0.f, 255.f);
        materialList.push_back(blackFog);
        geometryList.emplace_back(new ioVolumeBox(make_float3(0.f), b1size, 0.006f));
        Mat4 trans = ioTransform::translate(b1tr);
        trans *= ioTransform::rotateZ(z1Theta * (180.f / kPiF));
        trans *= ioTransform::rotateY(15.f);

        // white fog sphere, T(130,0,65)   (ioScene.h:741-749)
        Float3 b2origin = make_float3(165.f / 2.f, 75.f, 165.f / 2.f);
        Float3 b2tr = make_float3(130.f, 0.f, 65.f);
        materialList.push_back(whiteFog);
        geometryList.emplace_back(new ioVolumeSphere(b2origin.x, b2origin.y, b2origin.z, 75.f, 0.005f));
        Mat4 trans2 = ioTransform::translate(b2tr);

        identityInstances();
        geoInstList[6].setTransform(trans);
        geoInstList[7].setTransform(trans2);

        // no m_lightDefinitions.push_back: numLights 0, sky light on
        camera.reset(new ioPerspectiveCamera(278.f, 278.f, -800.f, 278.f, 278.f, 0.f, 0.0f, 1.0f, 0.0f, 40.0f,
                                             float(m_Nx) / float(m_Ny), /*aperture*/ 0.1f, /*focus_distance*/ 10.f));
    }

    int m_Nx = 0, m_Ny = 0, m_numSamples = 0, m_maxRayDepth = 0;
    std::string sceneDescription;
    std::vector<std::unique_ptr<ioTexture>> ownedTextures;
    std::vector<std::unique_ptr<ioMaterial>> ownedMaterials;
};

}  // namespace rtwhost
