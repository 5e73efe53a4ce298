/* A plain C program on the node entry points of libjpegdec_amd.so (include/jpegdec_amd.h, jda_node_*): one host process, every GPU.
 * Usage: node_user file.jpg n_images [n_devices | list of device ordinals "0,0"]
 * Decodes n_images copies of the file to RGB8888 -- sharded over the node's devices in contiguous blocks, pixels resident on the
 * device that decoded them -- and prints "devices D images N ok K checksum %016llx same S": K images with status 0, the checksum of
 * image 0's surface and whether all N checksums are equal.  Exit code 0, or the library's error (6 = no GPU: there is no CPU path). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jpegdec_amd.h"

int main(int argc, char **argv)
{
    if (argc < 3) return 100;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 101;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *jpeg = (uint8_t *)malloc((size_t)len);
    if (fread(jpeg, 1, (size_t)len, f) != (size_t)len) return 102;
    fclose(f);
    const int32_t n = atoi(argv[2]);
    int32_t want_dev = argc > 3 ? atoi(argv[3]) : 0, dev_list[16];
    const int32_t *devs = NULL;
    if (argc > 3 && strchr(argv[3], ',')) {                  /* a list of ordinals (one may come twice: two pipelines on one GPU) */
        want_dev = 0;
        for (char *t = strtok(argv[3], ","); t && want_dev < 16; t = strtok(NULL, ",")) dev_list[want_dev++] = atoi(t);
        devs = dev_list;
    }
    jda_image_info info;
    int rc = jda_parse(jpeg, (int32_t)len, &info);
    if (rc != JDA_SUCCESS) return rc;
    int32_t bpp, ow, oh, cw, ch, err = 0;
    rc = jda_output_geometry(&info, JDA_RGB8888, 0, &bpp, &ow, &oh, &cw, &ch);
    if (rc != JDA_SUCCESS) return rc;
    jda_node *node = jda_node_create(devs, want_dev, n, 2, 4, &err);
    if (!node) { printf("no node: error %d\n", err); return err ? err : 103; }
    const int32_t nd = jda_node_device_count(node);
    const int32_t pitch = (cw * bpp + 15) & ~15;
    const uint8_t **jpegs = (const uint8_t **)malloc(sizeof(*jpegs) * (size_t)n);
    int32_t *lens = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *pts = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *opts = (int32_t *)calloc((size_t)n, sizeof(int32_t)), *status = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *rowb = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    jda_output *outs = (jda_output *)malloc(sizeof(jda_output) * (size_t)n);
    uint64_t *sums = (uint64_t *)calloc((size_t)n, sizeof(uint64_t));
    void **blocks = (void **)calloc((size_t)nd, sizeof(void *));
    for (int32_t k = 0; k < nd; k++) {                       /* a device's block of the list lives in that device's memory */
        int32_t first, count;
        jda_node_shard(node, n, k, &first, &count);
        if (!count) continue;
        blocks[k] = jda_malloc(jda_node_context(node, k), (size_t)pitch * ch * count);
        if (!blocks[k]) return JDA_ERROR_MEMORY;
        for (int32_t i = 0; i < count; i++) {
            jda_output *o = &outs[first + i];
            o->pixels = (uint8_t *)blocks[k] + (size_t)i * pitch * ch; o->pitch_bytes = pitch; o->width_px = cw; o->rows = ch;
        }
    }
    for (int32_t i = 0; i < n; i++) { jpegs[i] = jpeg; lens[i] = (int32_t)len; pts[i] = JDA_RGB8888; rowb[i] = cw * bpp; status[i] = -1; }
    int32_t ticket = -1;
    rc = jda_node_submit(node, n, jpegs, lens, outs, pts, opts, &ticket);
    if (rc == JDA_SUCCESS) rc = jda_node_wait(node, ticket, status);
    if (rc == JDA_SUCCESS) rc = jda_node_checksums(node, n, outs, rowb, sums);
    int ok = 0, same = 1;
    for (int32_t i = 0; i < n; i++) { ok += status[i] == JDA_SUCCESS; same &= sums[i] == sums[0]; }
    /* the same list again from PAGE-LOCKED input (jda_host_alloc + JDA_SUBMIT_PINNED_INPUT: the copy engines read the file where it
     * lies, one file larger than the direct-copy threshold by repetition is not needed -- small files still take the mirror): the
     * same pixels must come out */
    if (rc == JDA_SUCCESS) {
        const uint64_t first_sum = sums[0];
        uint8_t *pinned = (uint8_t *)jda_host_alloc((size_t)len);
        if (!pinned) return JDA_ERROR_MEMORY;
        memcpy(pinned, jpeg, (size_t)len);
        for (int32_t i = 0; i < n; i++) { jpegs[i] = pinned; status[i] = -1; sums[i] = 0; }
        rc = jda_node_submit_ex(node, n, jpegs, lens, outs, pts, opts, JDA_SUBMIT_PINNED_INPUT, &ticket);
        if (rc == JDA_SUCCESS) rc = jda_node_wait(node, ticket, status);
        if (rc == JDA_SUCCESS) rc = jda_node_checksums(node, n, outs, rowb, sums);
        for (int32_t i = 0; i < n; i++) { same &= status[i] == JDA_SUCCESS && sums[i] == first_sum; }
        jda_host_free(pinned);
    }
    printf("devices %d images %d ok %d checksum %016llx same %d\n", nd, n, ok, (unsigned long long)sums[0], same);
    for (int32_t k = 0; k < nd; k++) if (blocks[k]) jda_free(jda_node_context(node, k), blocks[k]);
    jda_node_destroy(node);
    return rc;
}
