// stand-in for GLFW/glfw3.h (not vendored by the reference, not installed here): the types include/GlfwManager.h names and the
// calls src/RendererGUI.cpp makes (:49-105) -- for the syntax-only compile of tests/test_reference_gui_compiles.py; never linked
#pragma once
struct GLFWwindow;
struct GLFWmonitor;
struct GLFWvidmode;
extern "C" {
double glfwGetTime(void);
int glfwWindowShouldClose(GLFWwindow *window);
void glfwSwapBuffers(GLFWwindow *window);
void glfwPollEvents(void);
}
