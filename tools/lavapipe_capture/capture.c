/* Headless capture of the reference's draw on any Vulkan device (meant for Mesa lavapipe): the frame the reference
 * would present, read back to files, for the parity tests of this repository (tests/test_lavapipe_captures.py).
 *
 * It restates, with the plain Vulkan 1.1 C API and nothing else (no window, swap chain, vk-bootstrap or VMA):
 *   render pass and attachments      src/main.cpp:375-459, 723-738   (MSAA colour + depth + resolve; clear (0,0,0,1) / 1.0, :919-921)
 *   descriptor set layout            src/main.cpp:461-516            (binding 0 uniform block, 1-4 albedo/normal/metallic/roughness)
 *   graphics pipeline                src/main.cpp:518-655            (negative-height viewport, cull back, front CCW, LESS, no blend)
 *   recorded draw                    src/main.cpp:897-959            (vkCmdDrawIndexed(indices, 1, 0, 0, 0))
 *   texture image, mip chain, sampler src/texture.cpp:87-176, 178-271 (R8G8B8A8_SRGB, linear blits, LINEAR/LINEAR/REPEAT, maxAnisotropy)
 *   vertex layout                    src/mesh.hpp:18-54              (32-byte record: position, normal, uv)
 * and uses the reference's own shaders (shaders/shader.vert, shaders/shader.frag), compiled to SPIR-V by the Makefile
 * from where they lie.  Differences from the reference, none of which changes a pixel: the colour target is an
 * offscreen image in the swap chain's format (B8G8R8A8_SRGB, src/vulkan_helper/swapchain.cpp:17-21) whose final layout
 * is TRANSFER_SRC; the sample count is chosen by the caller instead of being the device's maximum
 * (src/vulkan_helper/gpu_device.cpp:16-30) so that both 1 and 4 can be captured; buffers are host-visible.
 *
 * A second, single-sample pass with the same vertex shader and a fragment shader that writes gl_PrimitiveID gives the
 * winning triangle of every pixel and the depth buffer (the reference itself never stores depth).
 *
 *   capture SCENE_DIR OUT_DIR        SCENE_DIR as written by export_scene.py; shader .spv files next to the binary
 *
 * This file has not been compiled by its author: the machine it was written on has no Vulkan headers.
 */
#include <vulkan/vulkan.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define VK(x) do { VkResult r_ = (x); if (r_ != VK_SUCCESS) { fprintf(stderr, "%s:%d: %s = %d\n", __FILE__, __LINE__, #x, (int)r_); exit(2); } } while (0)
#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); exit(2); } while (0)

static VkPhysicalDevice g_phys;
static VkDevice g_dev;
static VkQueue g_queue;
static VkCommandPool g_pool;
static VkPhysicalDeviceMemoryProperties g_mem;

static void* read_file(const char* dir, const char* name, size_t* size)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) DIE("cannot open %s", path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(n > 0 ? (size_t)n : 1);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) DIE("short read of %s", path);
    fclose(f);
    *size = (size_t)n;
    return p;
}
static void write_file(const char* dir, const char* name, const void* data, size_t size)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "wb");
    if (!f) DIE("cannot write %s", path);
    fwrite(data, 1, size, f);
    fclose(f);
}

static uint32_t memory_type(uint32_t bits, VkMemoryPropertyFlags want)
{
    for (uint32_t i = 0; i < g_mem.memoryTypeCount; ++i)
        if ((bits & (1u << i)) && (g_mem.memoryTypes[i].propertyFlags & want) == want) return i;
    DIE("no memory type with properties %x", (unsigned)want);
    return 0;
}

typedef struct { VkBuffer buf; VkDeviceMemory mem; void* map; VkDeviceSize size; } Buffer;
static Buffer make_buffer(VkDeviceSize size, VkBufferUsageFlags usage, const void* init)
{
    Buffer b = {0};
    b.size = size;
    const VkBufferCreateInfo ci = {.sType = VK_STRUCTURE_TYPE_BUFFER_CREATE_INFO, .size = size, .usage = usage, .sharingMode = VK_SHARING_MODE_EXCLUSIVE};
    VK(vkCreateBuffer(g_dev, &ci, NULL, &b.buf));
    VkMemoryRequirements req;
    vkGetBufferMemoryRequirements(g_dev, b.buf, &req);
    const VkMemoryAllocateInfo ai = {.sType = VK_STRUCTURE_TYPE_MEMORY_ALLOCATE_INFO, .allocationSize = req.size,
        .memoryTypeIndex = memory_type(req.memoryTypeBits, VK_MEMORY_PROPERTY_HOST_VISIBLE_BIT | VK_MEMORY_PROPERTY_HOST_COHERENT_BIT)};
    VK(vkAllocateMemory(g_dev, &ai, NULL, &b.mem));
    VK(vkBindBufferMemory(g_dev, b.buf, b.mem, 0));
    VK(vkMapMemory(g_dev, b.mem, 0, VK_WHOLE_SIZE, 0, &b.map));
    if (init) memcpy(b.map, init, (size_t)size);
    return b;
}

typedef struct { VkImage img; VkDeviceMemory mem; VkImageView view; uint32_t w, h, mips; VkFormat format; } Image;
static Image make_image(uint32_t w, uint32_t h, uint32_t mips, VkSampleCountFlagBits samples, VkFormat format, VkImageUsageFlags usage, VkImageAspectFlags aspect)
{
    Image im = {0};
    im.w = w; im.h = h; im.mips = mips; im.format = format;
    const VkImageCreateInfo ci = {.sType = VK_STRUCTURE_TYPE_IMAGE_CREATE_INFO, .imageType = VK_IMAGE_TYPE_2D, .format = format,
        .extent = {w, h, 1}, .mipLevels = mips, .arrayLayers = 1, .samples = samples, .tiling = VK_IMAGE_TILING_OPTIMAL, .usage = usage,
        .sharingMode = VK_SHARING_MODE_EXCLUSIVE, .initialLayout = VK_IMAGE_LAYOUT_UNDEFINED};
    VK(vkCreateImage(g_dev, &ci, NULL, &im.img));
    VkMemoryRequirements req;
    vkGetImageMemoryRequirements(g_dev, im.img, &req);
    const VkMemoryAllocateInfo ai = {.sType = VK_STRUCTURE_TYPE_MEMORY_ALLOCATE_INFO, .allocationSize = req.size,
        .memoryTypeIndex = memory_type(req.memoryTypeBits, VK_MEMORY_PROPERTY_DEVICE_LOCAL_BIT)};
    VK(vkAllocateMemory(g_dev, &ai, NULL, &im.mem));
    VK(vkBindImageMemory(g_dev, im.img, im.mem, 0));
    const VkImageViewCreateInfo vi = {.sType = VK_STRUCTURE_TYPE_IMAGE_VIEW_CREATE_INFO, .image = im.img, .viewType = VK_IMAGE_VIEW_TYPE_2D, .format = format,
        .subresourceRange = {.aspectMask = aspect, .baseMipLevel = 0, .levelCount = mips, .baseArrayLayer = 0, .layerCount = 1}};
    VK(vkCreateImageView(g_dev, &vi, NULL, &im.view));
    return im;
}

static VkCommandBuffer begin_once(void)
{
    const VkCommandBufferAllocateInfo ai = {.sType = VK_STRUCTURE_TYPE_COMMAND_BUFFER_ALLOCATE_INFO, .commandPool = g_pool,
        .level = VK_COMMAND_BUFFER_LEVEL_PRIMARY, .commandBufferCount = 1};
    VkCommandBuffer cb;
    VK(vkAllocateCommandBuffers(g_dev, &ai, &cb));
    const VkCommandBufferBeginInfo bi = {.sType = VK_STRUCTURE_TYPE_COMMAND_BUFFER_BEGIN_INFO, .flags = VK_COMMAND_BUFFER_USAGE_ONE_TIME_SUBMIT_BIT};
    VK(vkBeginCommandBuffer(cb, &bi));
    return cb;
}
static void end_once(VkCommandBuffer cb)
{
    VK(vkEndCommandBuffer(cb));
    const VkSubmitInfo si = {.sType = VK_STRUCTURE_TYPE_SUBMIT_INFO, .commandBufferCount = 1, .pCommandBuffers = &cb};
    VK(vkQueueSubmit(g_queue, 1, &si, VK_NULL_HANDLE));
    VK(vkQueueWaitIdle(g_queue));
    vkFreeCommandBuffers(g_dev, g_pool, 1, &cb);
}

static void image_barrier(VkCommandBuffer cb, VkImage img, VkImageAspectFlags aspect, uint32_t level, uint32_t levels, VkImageLayout from, VkImageLayout to,
                          VkAccessFlags src_access, VkAccessFlags dst_access, VkPipelineStageFlags src_stage, VkPipelineStageFlags dst_stage)
{
    const VkImageMemoryBarrier b = {.sType = VK_STRUCTURE_TYPE_IMAGE_MEMORY_BARRIER, .srcAccessMask = src_access, .dstAccessMask = dst_access,
        .oldLayout = from, .newLayout = to, .srcQueueFamilyIndex = VK_QUEUE_FAMILY_IGNORED, .dstQueueFamilyIndex = VK_QUEUE_FAMILY_IGNORED, .image = img,
        .subresourceRange = {.aspectMask = aspect, .baseMipLevel = level, .levelCount = levels, .baseArrayLayer = 0, .layerCount = 1}};
    vkCmdPipelineBarrier(cb, src_stage, dst_stage, 0, 0, NULL, 0, NULL, 1, &b);
}

/* create_texture + generate_mipmaps of the reference (src/texture.cpp:87-176, 178-271): level 0 from a staging buffer,
 * every further level a linear blit of the one before, all levels left in SHADER_READ_ONLY_OPTIMAL */
static Image make_texture(const void* rgba8, uint32_t w, uint32_t h, float max_anisotropy, VkBool32 anisotropy_supported, VkSampler* sampler)
{
    const uint32_t larger = w > h ? w : h;
    uint32_t mips = 1;
    while ((larger >> mips) > 0) ++mips;                     /* floor(log2(max(w, h))) + 1 */
    Image im = make_image(w, h, mips, VK_SAMPLE_COUNT_1_BIT, VK_FORMAT_R8G8B8A8_SRGB,
                          VK_IMAGE_USAGE_TRANSFER_SRC_BIT | VK_IMAGE_USAGE_TRANSFER_DST_BIT | VK_IMAGE_USAGE_SAMPLED_BIT, VK_IMAGE_ASPECT_COLOR_BIT);
    Buffer staging = make_buffer((VkDeviceSize)w * h * 4, VK_BUFFER_USAGE_TRANSFER_SRC_BIT, rgba8);
    VkCommandBuffer cb = begin_once();
    image_barrier(cb, im.img, VK_IMAGE_ASPECT_COLOR_BIT, 0, mips, VK_IMAGE_LAYOUT_UNDEFINED, VK_IMAGE_LAYOUT_TRANSFER_DST_OPTIMAL, 0, VK_ACCESS_TRANSFER_WRITE_BIT,
                  VK_PIPELINE_STAGE_TOP_OF_PIPE_BIT, VK_PIPELINE_STAGE_TRANSFER_BIT);
    const VkBufferImageCopy region = {.bufferOffset = 0, .bufferRowLength = 0, .bufferImageHeight = 0,
        .imageSubresource = {VK_IMAGE_ASPECT_COLOR_BIT, 0, 0, 1}, .imageOffset = {0, 0, 0}, .imageExtent = {w, h, 1}};
    vkCmdCopyBufferToImage(cb, staging.buf, im.img, VK_IMAGE_LAYOUT_TRANSFER_DST_OPTIMAL, 1, &region);
    int32_t mw = (int32_t)w, mh = (int32_t)h;
    for (uint32_t i = 1; i < mips; ++i) {
        image_barrier(cb, im.img, VK_IMAGE_ASPECT_COLOR_BIT, i - 1, 1, VK_IMAGE_LAYOUT_TRANSFER_DST_OPTIMAL, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL,
                      VK_ACCESS_TRANSFER_WRITE_BIT, VK_ACCESS_TRANSFER_READ_BIT, VK_PIPELINE_STAGE_TRANSFER_BIT, VK_PIPELINE_STAGE_TRANSFER_BIT);
        const int32_t nw = mw > 1 ? mw / 2 : 1, nh = mh > 1 ? mh / 2 : 1;
        const VkImageBlit blit = {.srcSubresource = {VK_IMAGE_ASPECT_COLOR_BIT, i - 1, 0, 1}, .srcOffsets = {{0, 0, 0}, {mw, mh, 1}},
                                  .dstSubresource = {VK_IMAGE_ASPECT_COLOR_BIT, i, 0, 1}, .dstOffsets = {{0, 0, 0}, {nw, nh, 1}}};
        vkCmdBlitImage(cb, im.img, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, im.img, VK_IMAGE_LAYOUT_TRANSFER_DST_OPTIMAL, 1, &blit, VK_FILTER_LINEAR);
        image_barrier(cb, im.img, VK_IMAGE_ASPECT_COLOR_BIT, i - 1, 1, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, VK_IMAGE_LAYOUT_SHADER_READ_ONLY_OPTIMAL,
                      VK_ACCESS_TRANSFER_READ_BIT, VK_ACCESS_SHADER_READ_BIT, VK_PIPELINE_STAGE_TRANSFER_BIT, VK_PIPELINE_STAGE_FRAGMENT_SHADER_BIT);
        mw = nw; mh = nh;
    }
    image_barrier(cb, im.img, VK_IMAGE_ASPECT_COLOR_BIT, mips - 1, 1, VK_IMAGE_LAYOUT_TRANSFER_DST_OPTIMAL, VK_IMAGE_LAYOUT_SHADER_READ_ONLY_OPTIMAL,
                  VK_ACCESS_TRANSFER_WRITE_BIT, VK_ACCESS_SHADER_READ_BIT, VK_PIPELINE_STAGE_TRANSFER_BIT, VK_PIPELINE_STAGE_FRAGMENT_SHADER_BIT);
    end_once(cb);
    vkUnmapMemory(g_dev, staging.mem);
    vkDestroyBuffer(g_dev, staging.buf, NULL);
    vkFreeMemory(g_dev, staging.mem, NULL);
    const VkBool32 aniso = anisotropy_supported && max_anisotropy > 1.0f;
    const VkSamplerCreateInfo si = {.sType = VK_STRUCTURE_TYPE_SAMPLER_CREATE_INFO, .magFilter = VK_FILTER_LINEAR, .minFilter = VK_FILTER_LINEAR,
        .mipmapMode = VK_SAMPLER_MIPMAP_MODE_LINEAR, .addressModeU = VK_SAMPLER_ADDRESS_MODE_REPEAT, .addressModeV = VK_SAMPLER_ADDRESS_MODE_REPEAT,
        .addressModeW = VK_SAMPLER_ADDRESS_MODE_REPEAT, .mipLodBias = 0.0f, .anisotropyEnable = aniso, .maxAnisotropy = aniso ? max_anisotropy : 1.0f,
        .compareEnable = VK_FALSE, .compareOp = VK_COMPARE_OP_ALWAYS, .minLod = 0.0f, .maxLod = (float)mips,
        .borderColor = VK_BORDER_COLOR_INT_OPAQUE_BLACK, .unnormalizedCoordinates = VK_FALSE};
    VK(vkCreateSampler(g_dev, &si, NULL, sampler));
    return im;
}

static VkShaderModule load_shader(const char* dir, const char* name)
{
    size_t size;
    void* code = read_file(dir, name, &size);
    const VkShaderModuleCreateInfo ci = {.sType = VK_STRUCTURE_TYPE_SHADER_MODULE_CREATE_INFO, .codeSize = size, .pCode = (const uint32_t*)code};
    VkShaderModule m;
    VK(vkCreateShaderModule(g_dev, &ci, NULL, &m));
    free(code);
    return m;
}

/* one render pass: colour (samples S) + depth (samples S) [+ single-sample resolve when S > 1]; the image that is read
 * back afterwards ends in TRANSFER_SRC_OPTIMAL, and so does the depth image when store_depth is set */
static VkRenderPass make_render_pass(VkFormat color_format, VkSampleCountFlagBits samples, VkBool32 store_depth)
{
    const VkBool32 msaa = samples != VK_SAMPLE_COUNT_1_BIT;
    VkAttachmentDescription att[3];
    memset(att, 0, sizeof att);
    att[0] = (VkAttachmentDescription){.format = color_format, .samples = samples, .loadOp = VK_ATTACHMENT_LOAD_OP_CLEAR, .storeOp = VK_ATTACHMENT_STORE_OP_STORE,
        .stencilLoadOp = VK_ATTACHMENT_LOAD_OP_DONT_CARE, .stencilStoreOp = VK_ATTACHMENT_STORE_OP_DONT_CARE, .initialLayout = VK_IMAGE_LAYOUT_UNDEFINED,
        .finalLayout = msaa ? VK_IMAGE_LAYOUT_COLOR_ATTACHMENT_OPTIMAL : VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL};
    att[1] = (VkAttachmentDescription){.format = VK_FORMAT_D32_SFLOAT, .samples = samples, .loadOp = VK_ATTACHMENT_LOAD_OP_CLEAR,
        .storeOp = store_depth ? VK_ATTACHMENT_STORE_OP_STORE : VK_ATTACHMENT_STORE_OP_DONT_CARE,
        .stencilLoadOp = VK_ATTACHMENT_LOAD_OP_DONT_CARE, .stencilStoreOp = VK_ATTACHMENT_STORE_OP_DONT_CARE, .initialLayout = VK_IMAGE_LAYOUT_UNDEFINED,
        .finalLayout = store_depth ? VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL : VK_IMAGE_LAYOUT_DEPTH_STENCIL_ATTACHMENT_OPTIMAL};
    att[2] = (VkAttachmentDescription){.format = color_format, .samples = VK_SAMPLE_COUNT_1_BIT, .loadOp = VK_ATTACHMENT_LOAD_OP_DONT_CARE,
        .storeOp = VK_ATTACHMENT_STORE_OP_STORE, .stencilLoadOp = VK_ATTACHMENT_LOAD_OP_DONT_CARE, .stencilStoreOp = VK_ATTACHMENT_STORE_OP_DONT_CARE,
        .initialLayout = VK_IMAGE_LAYOUT_UNDEFINED, .finalLayout = VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL};
    const VkAttachmentReference color_ref = {0, VK_IMAGE_LAYOUT_COLOR_ATTACHMENT_OPTIMAL};
    const VkAttachmentReference depth_ref = {1, VK_IMAGE_LAYOUT_DEPTH_STENCIL_ATTACHMENT_OPTIMAL};
    const VkAttachmentReference resolve_ref = {2, VK_IMAGE_LAYOUT_COLOR_ATTACHMENT_OPTIMAL};
    const VkSubpassDescription subpass = {.pipelineBindPoint = VK_PIPELINE_BIND_POINT_GRAPHICS, .colorAttachmentCount = 1, .pColorAttachments = &color_ref,
        .pResolveAttachments = msaa ? &resolve_ref : NULL, .pDepthStencilAttachment = &depth_ref};
    const VkSubpassDependency deps[2] = {
        {.srcSubpass = VK_SUBPASS_EXTERNAL, .dstSubpass = 0,
         .srcStageMask = VK_PIPELINE_STAGE_COLOR_ATTACHMENT_OUTPUT_BIT | VK_PIPELINE_STAGE_EARLY_FRAGMENT_TESTS_BIT,
         .dstStageMask = VK_PIPELINE_STAGE_COLOR_ATTACHMENT_OUTPUT_BIT | VK_PIPELINE_STAGE_EARLY_FRAGMENT_TESTS_BIT, .srcAccessMask = 0,
         .dstAccessMask = VK_ACCESS_COLOR_ATTACHMENT_WRITE_BIT | VK_ACCESS_DEPTH_STENCIL_ATTACHMENT_WRITE_BIT},
        {.srcSubpass = 0, .dstSubpass = VK_SUBPASS_EXTERNAL,
         .srcStageMask = VK_PIPELINE_STAGE_COLOR_ATTACHMENT_OUTPUT_BIT | VK_PIPELINE_STAGE_LATE_FRAGMENT_TESTS_BIT, .dstStageMask = VK_PIPELINE_STAGE_TRANSFER_BIT,
         .srcAccessMask = VK_ACCESS_COLOR_ATTACHMENT_WRITE_BIT | VK_ACCESS_DEPTH_STENCIL_ATTACHMENT_WRITE_BIT, .dstAccessMask = VK_ACCESS_TRANSFER_READ_BIT}};
    const VkRenderPassCreateInfo ci = {.sType = VK_STRUCTURE_TYPE_RENDER_PASS_CREATE_INFO, .attachmentCount = msaa ? 3u : 2u, .pAttachments = att,
        .subpassCount = 1, .pSubpasses = &subpass, .dependencyCount = 2, .pDependencies = deps};
    VkRenderPass rp;
    VK(vkCreateRenderPass(g_dev, &ci, NULL, &rp));
    return rp;
}

static VkPipeline make_pipeline(VkRenderPass rp, VkPipelineLayout layout, VkShaderModule vert, VkShaderModule frag, uint32_t W, uint32_t H, VkSampleCountFlagBits samples)
{
    const VkPipelineShaderStageCreateInfo stages[2] = {
        {.sType = VK_STRUCTURE_TYPE_PIPELINE_SHADER_STAGE_CREATE_INFO, .stage = VK_SHADER_STAGE_VERTEX_BIT, .module = vert, .pName = "main"},
        {.sType = VK_STRUCTURE_TYPE_PIPELINE_SHADER_STAGE_CREATE_INFO, .stage = VK_SHADER_STAGE_FRAGMENT_BIT, .module = frag, .pName = "main"}};
    const VkVertexInputBindingDescription binding = {.binding = 0, .stride = 32, .inputRate = VK_VERTEX_INPUT_RATE_VERTEX};
    const VkVertexInputAttributeDescription attrs[3] = {{.location = 0, .binding = 0, .format = VK_FORMAT_R32G32B32_SFLOAT, .offset = 0},
                                                        {.location = 1, .binding = 0, .format = VK_FORMAT_R32G32B32_SFLOAT, .offset = 12},
                                                        {.location = 2, .binding = 0, .format = VK_FORMAT_R32G32_SFLOAT, .offset = 24}};
    const VkPipelineVertexInputStateCreateInfo vertex_input = {.sType = VK_STRUCTURE_TYPE_PIPELINE_VERTEX_INPUT_STATE_CREATE_INFO,
        .vertexBindingDescriptionCount = 1, .pVertexBindingDescriptions = &binding, .vertexAttributeDescriptionCount = 3, .pVertexAttributeDescriptions = attrs};
    const VkPipelineInputAssemblyStateCreateInfo assembly = {.sType = VK_STRUCTURE_TYPE_PIPELINE_INPUT_ASSEMBLY_STATE_CREATE_INFO,
        .topology = VK_PRIMITIVE_TOPOLOGY_TRIANGLE_LIST, .primitiveRestartEnable = VK_FALSE};
    const VkViewport viewport = {.x = 0.0f, .y = (float)H, .width = (float)W, .height = -(float)H, .minDepth = 0.0f, .maxDepth = 1.0f};
    const VkRect2D scissor = {{0, 0}, {W, H}};
    const VkPipelineViewportStateCreateInfo viewport_state = {.sType = VK_STRUCTURE_TYPE_PIPELINE_VIEWPORT_STATE_CREATE_INFO, .viewportCount = 1,
        .pViewports = &viewport, .scissorCount = 1, .pScissors = &scissor};
    const VkPipelineRasterizationStateCreateInfo raster = {.sType = VK_STRUCTURE_TYPE_PIPELINE_RASTERIZATION_STATE_CREATE_INFO, .depthClampEnable = VK_FALSE,
        .rasterizerDiscardEnable = VK_FALSE, .polygonMode = VK_POLYGON_MODE_FILL, .cullMode = VK_CULL_MODE_BACK_BIT,
        .frontFace = VK_FRONT_FACE_COUNTER_CLOCKWISE, .depthBiasEnable = VK_FALSE, .lineWidth = 1.0f};
    const VkPipelineMultisampleStateCreateInfo multisample = {.sType = VK_STRUCTURE_TYPE_PIPELINE_MULTISAMPLE_STATE_CREATE_INFO,
        .rasterizationSamples = samples, .sampleShadingEnable = VK_FALSE};
    const VkPipelineDepthStencilStateCreateInfo depth = {.sType = VK_STRUCTURE_TYPE_PIPELINE_DEPTH_STENCIL_STATE_CREATE_INFO, .depthTestEnable = VK_TRUE,
        .depthWriteEnable = VK_TRUE, .depthCompareOp = VK_COMPARE_OP_LESS, .depthBoundsTestEnable = VK_FALSE, .stencilTestEnable = VK_FALSE};
    const VkPipelineColorBlendAttachmentState blend_att = {.blendEnable = VK_FALSE,
        .colorWriteMask = VK_COLOR_COMPONENT_R_BIT | VK_COLOR_COMPONENT_G_BIT | VK_COLOR_COMPONENT_B_BIT | VK_COLOR_COMPONENT_A_BIT};
    const VkPipelineColorBlendStateCreateInfo blend = {.sType = VK_STRUCTURE_TYPE_PIPELINE_COLOR_BLEND_STATE_CREATE_INFO, .logicOpEnable = VK_FALSE,
        .logicOp = VK_LOGIC_OP_COPY, .attachmentCount = 1, .pAttachments = &blend_att};
    const VkGraphicsPipelineCreateInfo ci = {.sType = VK_STRUCTURE_TYPE_GRAPHICS_PIPELINE_CREATE_INFO, .stageCount = 2, .pStages = stages,
        .pVertexInputState = &vertex_input, .pInputAssemblyState = &assembly, .pViewportState = &viewport_state, .pRasterizationState = &raster,
        .pMultisampleState = &multisample, .pDepthStencilState = &depth, .pColorBlendState = &blend, .layout = layout, .renderPass = rp, .subpass = 0};
    VkPipeline p;
    VK(vkCreateGraphicsPipelines(g_dev, VK_NULL_HANDLE, 1, &ci, NULL, &p));
    return p;
}

typedef struct { uint32_t width, height, samples, n_vertices, n_indices, tex_w[4], tex_h[4]; float max_anisotropy; } Scene;
static Scene read_scene(const char* dir)
{
    Scene s;
    memset(&s, 0, sizeof s);
    s.samples = 1; s.max_anisotropy = 16.0f;
    char path[4096], key[64];
    snprintf(path, sizeof path, "%s/scene.txt", dir);
    FILE* f = fopen(path, "r");
    if (!f) DIE("cannot open %s", path);
    while (fscanf(f, "%63s", key) == 1) {
        if (!strcmp(key, "width")) fscanf(f, "%u", &s.width);
        else if (!strcmp(key, "height")) fscanf(f, "%u", &s.height);
        else if (!strcmp(key, "samples")) fscanf(f, "%u", &s.samples);
        else if (!strcmp(key, "max_anisotropy")) fscanf(f, "%f", &s.max_anisotropy);
        else if (!strcmp(key, "n_vertices")) fscanf(f, "%u", &s.n_vertices);
        else if (!strcmp(key, "n_indices")) fscanf(f, "%u", &s.n_indices);
        else if (!strncmp(key, "tex", 3) && key[3] >= '0' && key[3] <= '3') fscanf(f, "%u %u", &s.tex_w[key[3] - '0'], &s.tex_h[key[3] - '0']);
        else DIE("scene.txt: unknown key %s", key);
    }
    fclose(f);
    if (!s.width || !s.height || !s.n_vertices || !s.n_indices) DIE("scene.txt is incomplete");
    if (s.samples != 1 && s.samples != 2 && s.samples != 4 && s.samples != 8) DIE("samples must be 1, 2, 4 or 8");
    return s;
}

int main(int argc, char** argv)
{
    if (argc < 3) DIE("usage: capture SCENE_DIR OUT_DIR   (shader.vert.spv, shader.frag.spv, ids.frag.spv next to the working directory or in $CAPTURE_SHADER_DIR)");
    const char* scene_dir = argv[1];
    const char* out_dir = argv[2];
    const char* shader_dir = getenv("CAPTURE_SHADER_DIR") ? getenv("CAPTURE_SHADER_DIR") : ".";
    const Scene sc = read_scene(scene_dir);
    const uint32_t W = sc.width, H = sc.height;

    /* instance, device (CAPTURE_DEVICE=n picks the n-th physical device; default: the first whose name contains "llvmpipe", else the first) */
    const VkApplicationInfo app = {.sType = VK_STRUCTURE_TYPE_APPLICATION_INFO, .pApplicationName = "lavapipe_capture", .apiVersion = VK_API_VERSION_1_1};
    const VkInstanceCreateInfo ici = {.sType = VK_STRUCTURE_TYPE_INSTANCE_CREATE_INFO, .pApplicationInfo = &app};
    VkInstance instance;
    VK(vkCreateInstance(&ici, NULL, &instance));
    uint32_t n_phys = 0;
    VK(vkEnumeratePhysicalDevices(instance, &n_phys, NULL));
    if (!n_phys) DIE("no Vulkan device");
    VkPhysicalDevice* phys = malloc(n_phys * sizeof *phys);
    VK(vkEnumeratePhysicalDevices(instance, &n_phys, phys));
    uint32_t pick = 0;
    if (getenv("CAPTURE_DEVICE")) pick = (uint32_t)atoi(getenv("CAPTURE_DEVICE"));
    else
        for (uint32_t i = 0; i < n_phys; ++i) {
            VkPhysicalDeviceProperties p;
            vkGetPhysicalDeviceProperties(phys[i], &p);
            if (strstr(p.deviceName, "llvmpipe")) { pick = i; break; }
        }
    if (pick >= n_phys) DIE("CAPTURE_DEVICE out of range");
    g_phys = phys[pick];
    VkPhysicalDeviceProperties props;
    vkGetPhysicalDeviceProperties(g_phys, &props);
    VkPhysicalDeviceFeatures have;
    vkGetPhysicalDeviceFeatures(g_phys, &have);
    vkGetPhysicalDeviceMemoryProperties(g_phys, &g_mem);
    uint32_t n_fam = 0, family = UINT32_MAX;
    vkGetPhysicalDeviceQueueFamilyProperties(g_phys, &n_fam, NULL);
    VkQueueFamilyProperties* fams = malloc(n_fam * sizeof *fams);
    vkGetPhysicalDeviceQueueFamilyProperties(g_phys, &n_fam, fams);
    for (uint32_t i = 0; i < n_fam; ++i)
        if (fams[i].queueFlags & VK_QUEUE_GRAPHICS_BIT) { family = i; break; }
    if (family == UINT32_MAX) DIE("no graphics queue");
    const float prio = 1.0f;
    const VkDeviceQueueCreateInfo qci = {.sType = VK_STRUCTURE_TYPE_DEVICE_QUEUE_CREATE_INFO, .queueFamilyIndex = family, .queueCount = 1, .pQueuePriorities = &prio};
    VkPhysicalDeviceFeatures want;
    memset(&want, 0, sizeof want);
    want.samplerAnisotropy = have.samplerAnisotropy;     /* the reference asks for it (src/vulkan_helper/gpu_device.cpp) */
    want.geometryShader = have.geometryShader;           /* gl_PrimitiveID in a fragment shader needs the Geometry capability */
    const VkDeviceCreateInfo dci = {.sType = VK_STRUCTURE_TYPE_DEVICE_CREATE_INFO, .queueCreateInfoCount = 1, .pQueueCreateInfos = &qci, .pEnabledFeatures = &want};
    VK(vkCreateDevice(g_phys, &dci, NULL, &g_dev));
    vkGetDeviceQueue(g_dev, family, 0, &g_queue);
    const VkCommandPoolCreateInfo pci = {.sType = VK_STRUCTURE_TYPE_COMMAND_POOL_CREATE_INFO, .queueFamilyIndex = family};
    VK(vkCreateCommandPool(g_dev, &pci, NULL, &g_pool));
    const VkSampleCountFlags counts = props.limits.framebufferColorSampleCounts & props.limits.framebufferDepthSampleCounts;
    if (!(counts & sc.samples)) DIE("the device does not render %u samples (mask %x)", sc.samples, (unsigned)counts);
    const VkSampleCountFlagBits samples = (VkSampleCountFlagBits)sc.samples;

    /* scene data */
    size_t size;
    void* vertices = read_file(scene_dir, "vertices.bin", &size);
    if (size != (size_t)sc.n_vertices * 32) DIE("vertices.bin: %zu bytes for %u vertices", size, sc.n_vertices);
    void* indices = read_file(scene_dir, "indices.bin", &size);
    if (size != (size_t)sc.n_indices * 4) DIE("indices.bin: %zu bytes for %u indices", size, sc.n_indices);
    void* ubo = read_file(scene_dir, "ubo.bin", &size);
    if (size != 208) DIE("ubo.bin must hold the 208-byte std140 block");
    Buffer vb = make_buffer((VkDeviceSize)sc.n_vertices * 32, VK_BUFFER_USAGE_VERTEX_BUFFER_BIT, vertices);
    Buffer ib = make_buffer((VkDeviceSize)sc.n_indices * 4, VK_BUFFER_USAGE_INDEX_BUFFER_BIT, indices);
    Buffer ub = make_buffer(208, VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT, ubo);
    Image tex[4];
    VkSampler samplers[4];
    for (int i = 0; i < 4; ++i) {
        char name[32];
        snprintf(name, sizeof name, "tex%d.rgba", i);
        void* texels = read_file(scene_dir, name, &size);
        if (size != (size_t)sc.tex_w[i] * sc.tex_h[i] * 4) DIE("%s: %zu bytes for %ux%u", name, size, sc.tex_w[i], sc.tex_h[i]);
        tex[i] = make_texture(texels, sc.tex_w[i], sc.tex_h[i], sc.max_anisotropy, have.samplerAnisotropy, &samplers[i]);
        free(texels);
    }

    /* descriptor set (src/main.cpp:461-516, 755-895) */
    VkDescriptorSetLayoutBinding bindings[5];
    memset(bindings, 0, sizeof bindings);
    bindings[0] = (VkDescriptorSetLayoutBinding){.binding = 0, .descriptorType = VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER, .descriptorCount = 1, .stageFlags = VK_SHADER_STAGE_VERTEX_BIT};
    for (uint32_t i = 1; i <= 4; ++i)
        bindings[i] = (VkDescriptorSetLayoutBinding){.binding = i, .descriptorType = VK_DESCRIPTOR_TYPE_COMBINED_IMAGE_SAMPLER, .descriptorCount = 1,
                                                     .stageFlags = VK_SHADER_STAGE_FRAGMENT_BIT};
    const VkDescriptorSetLayoutCreateInfo lci = {.sType = VK_STRUCTURE_TYPE_DESCRIPTOR_SET_LAYOUT_CREATE_INFO, .bindingCount = 5, .pBindings = bindings};
    VkDescriptorSetLayout set_layout;
    VK(vkCreateDescriptorSetLayout(g_dev, &lci, NULL, &set_layout));
    const VkDescriptorPoolSize pool_sizes[2] = {{VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER, 1}, {VK_DESCRIPTOR_TYPE_COMBINED_IMAGE_SAMPLER, 4}};
    const VkDescriptorPoolCreateInfo dpci = {.sType = VK_STRUCTURE_TYPE_DESCRIPTOR_POOL_CREATE_INFO, .maxSets = 1, .poolSizeCount = 2, .pPoolSizes = pool_sizes};
    VkDescriptorPool dpool;
    VK(vkCreateDescriptorPool(g_dev, &dpci, NULL, &dpool));
    const VkDescriptorSetAllocateInfo dsai = {.sType = VK_STRUCTURE_TYPE_DESCRIPTOR_SET_ALLOCATE_INFO, .descriptorPool = dpool, .descriptorSetCount = 1, .pSetLayouts = &set_layout};
    VkDescriptorSet dset;
    VK(vkAllocateDescriptorSets(g_dev, &dsai, &dset));
    const VkDescriptorBufferInfo ubo_info = {ub.buf, 0, 208};
    VkDescriptorImageInfo image_infos[4];
    VkWriteDescriptorSet writes[5];
    memset(writes, 0, sizeof writes);
    writes[0] = (VkWriteDescriptorSet){.sType = VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET, .dstSet = dset, .dstBinding = 0, .descriptorCount = 1,
                                       .descriptorType = VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER, .pBufferInfo = &ubo_info};
    for (uint32_t i = 0; i < 4; ++i) {
        image_infos[i] = (VkDescriptorImageInfo){samplers[i], tex[i].view, VK_IMAGE_LAYOUT_SHADER_READ_ONLY_OPTIMAL};
        writes[i + 1] = (VkWriteDescriptorSet){.sType = VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET, .dstSet = dset, .dstBinding = i + 1, .descriptorCount = 1,
                                               .descriptorType = VK_DESCRIPTOR_TYPE_COMBINED_IMAGE_SAMPLER, .pImageInfo = &image_infos[i]};
    }
    vkUpdateDescriptorSets(g_dev, 5, writes, 0, NULL);
    const VkPipelineLayoutCreateInfo plci = {.sType = VK_STRUCTURE_TYPE_PIPELINE_LAYOUT_CREATE_INFO, .setLayoutCount = 1, .pSetLayouts = &set_layout};
    VkPipelineLayout layout;
    VK(vkCreatePipelineLayout(g_dev, &plci, NULL, &layout));

    /* targets: the frame (samples S, resolved when S > 1) and the single-sample id pass */
    const VkFormat color_format = VK_FORMAT_B8G8R8A8_SRGB;
    const VkBool32 msaa = samples != VK_SAMPLE_COUNT_1_BIT;
    Image color = make_image(W, H, 1, samples, color_format, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT | VK_IMAGE_USAGE_TRANSFER_SRC_BIT, VK_IMAGE_ASPECT_COLOR_BIT);
    Image depth = make_image(W, H, 1, samples, VK_FORMAT_D32_SFLOAT, VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT, VK_IMAGE_ASPECT_DEPTH_BIT);
    Image resolve = color;
    if (msaa) resolve = make_image(W, H, 1, VK_SAMPLE_COUNT_1_BIT, color_format, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT | VK_IMAGE_USAGE_TRANSFER_SRC_BIT, VK_IMAGE_ASPECT_COLOR_BIT);
    Image ids = make_image(W, H, 1, VK_SAMPLE_COUNT_1_BIT, VK_FORMAT_R32_UINT, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT | VK_IMAGE_USAGE_TRANSFER_SRC_BIT, VK_IMAGE_ASPECT_COLOR_BIT);
    Image ids_depth = make_image(W, H, 1, VK_SAMPLE_COUNT_1_BIT, VK_FORMAT_D32_SFLOAT, VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT | VK_IMAGE_USAGE_TRANSFER_SRC_BIT,
                                 VK_IMAGE_ASPECT_DEPTH_BIT);
    VkRenderPass frame_pass = make_render_pass(color_format, samples, VK_FALSE);
    VkRenderPass ids_pass = make_render_pass(VK_FORMAT_R32_UINT, VK_SAMPLE_COUNT_1_BIT, VK_TRUE);
    const VkImageView frame_views[3] = {color.view, depth.view, resolve.view};
    const VkFramebufferCreateInfo ffci = {.sType = VK_STRUCTURE_TYPE_FRAMEBUFFER_CREATE_INFO, .renderPass = frame_pass, .attachmentCount = msaa ? 3u : 2u,
        .pAttachments = frame_views, .width = W, .height = H, .layers = 1};
    VkFramebuffer frame_fb;
    VK(vkCreateFramebuffer(g_dev, &ffci, NULL, &frame_fb));
    const VkImageView ids_views[2] = {ids.view, ids_depth.view};
    const VkFramebufferCreateInfo ifci = {.sType = VK_STRUCTURE_TYPE_FRAMEBUFFER_CREATE_INFO, .renderPass = ids_pass, .attachmentCount = 2,
        .pAttachments = ids_views, .width = W, .height = H, .layers = 1};
    VkFramebuffer ids_fb;
    VK(vkCreateFramebuffer(g_dev, &ifci, NULL, &ids_fb));

    VkShaderModule vert = load_shader(shader_dir, "shader.vert.spv"), frag = load_shader(shader_dir, "shader.frag.spv");
    VkPipeline frame_pipeline = make_pipeline(frame_pass, layout, vert, frag, W, H, samples);
    VkPipeline ids_pipeline = VK_NULL_HANDLE;
    if (have.geometryShader) ids_pipeline = make_pipeline(ids_pass, layout, vert, load_shader(shader_dir, "ids.frag.spv"), W, H, VK_SAMPLE_COUNT_1_BIT);

    Buffer color_out = make_buffer((VkDeviceSize)W * H * 4, VK_BUFFER_USAGE_TRANSFER_DST_BIT, NULL);
    Buffer ids_out = make_buffer((VkDeviceSize)W * H * 4, VK_BUFFER_USAGE_TRANSFER_DST_BIT, NULL);
    Buffer depth_out = make_buffer((VkDeviceSize)W * H * 4, VK_BUFFER_USAGE_TRANSFER_DST_BIT, NULL);

    /* the recorded frame (src/main.cpp:897-959) and its read-back */
    VkCommandBuffer cb = begin_once();
    VkClearValue clears[3];
    memset(clears, 0, sizeof clears);
    clears[0].color.float32[3] = 1.0f;                       /* {0, 0, 0, 1} */
    clears[1].depthStencil.depth = 1.0f;
    const VkRenderPassBeginInfo fbi = {.sType = VK_STRUCTURE_TYPE_RENDER_PASS_BEGIN_INFO, .renderPass = frame_pass, .framebuffer = frame_fb,
        .renderArea = {{0, 0}, {W, H}}, .clearValueCount = msaa ? 3u : 2u, .pClearValues = clears};
    const VkDeviceSize zero = 0;
    vkCmdBeginRenderPass(cb, &fbi, VK_SUBPASS_CONTENTS_INLINE);
    vkCmdBindPipeline(cb, VK_PIPELINE_BIND_POINT_GRAPHICS, frame_pipeline);
    vkCmdBindVertexBuffers(cb, 0, 1, &vb.buf, &zero);
    vkCmdBindIndexBuffer(cb, ib.buf, 0, VK_INDEX_TYPE_UINT32);
    vkCmdBindDescriptorSets(cb, VK_PIPELINE_BIND_POINT_GRAPHICS, layout, 0, 1, &dset, 0, NULL);
    vkCmdDrawIndexed(cb, sc.n_indices, 1, 0, 0, 0);
    vkCmdEndRenderPass(cb);
    const VkBufferImageCopy whole_color = {.imageSubresource = {VK_IMAGE_ASPECT_COLOR_BIT, 0, 0, 1}, .imageExtent = {W, H, 1}};
    vkCmdCopyImageToBuffer(cb, resolve.img, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, color_out.buf, 1, &whole_color);
    if (ids_pipeline) {
        VkClearValue id_clears[2];
        memset(id_clears, 0, sizeof id_clears);
        id_clears[0].color.uint32[0] = 0xFFFFFFFFu;          /* no triangle */
        id_clears[1].depthStencil.depth = 1.0f;
        const VkRenderPassBeginInfo ibi = {.sType = VK_STRUCTURE_TYPE_RENDER_PASS_BEGIN_INFO, .renderPass = ids_pass, .framebuffer = ids_fb,
            .renderArea = {{0, 0}, {W, H}}, .clearValueCount = 2, .pClearValues = id_clears};
        vkCmdBeginRenderPass(cb, &ibi, VK_SUBPASS_CONTENTS_INLINE);
        vkCmdBindPipeline(cb, VK_PIPELINE_BIND_POINT_GRAPHICS, ids_pipeline);
        vkCmdBindVertexBuffers(cb, 0, 1, &vb.buf, &zero);
        vkCmdBindIndexBuffer(cb, ib.buf, 0, VK_INDEX_TYPE_UINT32);
        vkCmdBindDescriptorSets(cb, VK_PIPELINE_BIND_POINT_GRAPHICS, layout, 0, 1, &dset, 0, NULL);
        vkCmdDrawIndexed(cb, sc.n_indices, 1, 0, 0, 0);
        vkCmdEndRenderPass(cb);
        vkCmdCopyImageToBuffer(cb, ids.img, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, ids_out.buf, 1, &whole_color);
        const VkBufferImageCopy whole_depth = {.imageSubresource = {VK_IMAGE_ASPECT_DEPTH_BIT, 0, 0, 1}, .imageExtent = {W, H, 1}};
        vkCmdCopyImageToBuffer(cb, ids_depth.img, VK_IMAGE_LAYOUT_TRANSFER_SRC_OPTIMAL, depth_out.buf, 1, &whole_depth);
    }
    const VkMemoryBarrier to_host = {.sType = VK_STRUCTURE_TYPE_MEMORY_BARRIER, .srcAccessMask = VK_ACCESS_TRANSFER_WRITE_BIT, .dstAccessMask = VK_ACCESS_HOST_READ_BIT};
    vkCmdPipelineBarrier(cb, VK_PIPELINE_STAGE_TRANSFER_BIT, VK_PIPELINE_STAGE_HOST_BIT, 0, 1, &to_host, 0, NULL, 0, NULL);
    end_once(cb);

    write_file(out_dir, "color_bgra8.bin", color_out.map, (size_t)W * H * 4);
    if (ids_pipeline) {
        write_file(out_dir, "ids_u32.bin", ids_out.map, (size_t)W * H * 4);
        write_file(out_dir, "depth_f32.bin", depth_out.map, (size_t)W * H * 4);
    }
    char meta[2048];
    const int n = snprintf(meta, sizeof meta,
        "device %s\napi_version %u.%u.%u\ndriver_version %u\nvendor_id %u\nwidth %u\nheight %u\nsamples %u\nmax_anisotropy %g\n"
        "sampler_anisotropy_supported %u\nmax_sampler_anisotropy %g\nsub_pixel_precision_bits %u\nsub_texel_precision_bits %u\nmipmap_precision_bits %u\n"
        "standard_sample_locations %u\nframebuffer_sample_counts %u\nids_and_depth %u\n",
        props.deviceName, VK_VERSION_MAJOR(props.apiVersion), VK_VERSION_MINOR(props.apiVersion), VK_VERSION_PATCH(props.apiVersion), props.driverVersion,
        props.vendorID, W, H, sc.samples, (double)sc.max_anisotropy, (unsigned)have.samplerAnisotropy, (double)props.limits.maxSamplerAnisotropy,
        props.limits.subPixelPrecisionBits, props.limits.subTexelPrecisionBits, props.limits.mipmapPrecisionBits,
        (unsigned)props.limits.standardSampleLocations, (unsigned)counts, ids_pipeline ? 1u : 0u);
    write_file(out_dir, "meta.txt", meta, (size_t)n);
    printf("captured %ux%u, %u sample(s), on %s\n", W, H, sc.samples, props.deviceName);
    vkDeviceWaitIdle(g_dev);
    return 0;   /* the process ends here: the driver reclaims every object */
}
